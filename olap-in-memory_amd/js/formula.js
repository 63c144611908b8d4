'use strict';
/*
 * formula.js — computed-measure formulas ("next" row f4 of SURVEY.md §8(f)).
 *
 * The reference parses and evaluates formulas with the third-party `@growblocks/expr-eval`
 * (git dependency, package.json:11; configured in src/parser.js:3-26 with logical / comparison /
 * `in` / assignment operators switched OFF, `isNaN` added, and `||` re-purposed as a NaN-coalescing
 * addition).  That module is absent offline, so this is an independent implementation of the
 * arithmetic subset the configuration leaves enabled:
 *
 *   numbers, identifiers (measure ids, `<id>__total`), constants PI E true false
 *   a ? b : c         conditional (lowest precedence, right-associative)
 *   + - ||            additive level, left-associative ( a || b : the non-NaN one, else a + b )
 *   * / %             multiplicative
 *   - + and named prefix functions (abs ceil floor round trunc sqrt cbrt exp ln log log10 log2 sign
 *                     sin cos tan asin acos atan not) with or without parentheses
 *   ^                 power, right-associative
 *   f(a, b, ...)      min max pow atan2 hypot if roundTo isNaN
 *
 * PARITY: pinned only by the reference tests' literals (test/cube-accessors.js:58-67,
 * test/cube-to-cube.js:360-382); everything else is "parity unpinned" (the evaluator lives in a
 * module that is not in /root/reference).
 *
 * An Expression offers what src/cube.js uses: variables({withMembers}), evaluate(params),
 * toString() (fully parenthesised, re-parsable), substitute(name, replacement) — and compile(),
 * which lowers the tree to the postfix program the HIP element-wise interpreter runs
 * (olap_eval_formula, include/olap_hip.h).
 */

const UNARY = ['abs', 'ceil', 'floor', 'round', 'trunc', 'sqrt', 'cbrt', 'exp', 'ln', 'log', 'log10', 'log2', 'sign', 'sin', 'cos', 'tan', 'asin', 'acos', 'atan', 'not'];
const CALLS = { min: -1, max: -1, pow: 2, atan2: 2, hypot: -1, if: 3, roundTo: 2, isNaN: 1 };
const CONSTANTS = { PI: Math.PI, E: Math.E, true: 1, false: 0 };

// opcodes shared with the device interpreter (olap_kernels.hpp: FormulaOp) — keep in sync
const OP = {
  CONST: 0, INPUT: 1, SCALAR: 2, ADD: 3, SUB: 4, MUL: 5, DIV: 6, MOD: 7, POW: 8, NEG: 9, NANADD: 10, SELECT: 11,
  MIN: 12, MAX: 13, ATAN2: 14, HYPOT: 15, ROUNDTO: 16, ISNAN: 17,
  ABS: 20, CEIL: 21, FLOOR: 22, ROUND: 23, TRUNC: 24, SQRT: 25, CBRT: 26, EXP: 27, LN: 28, LOG10: 29, LOG2: 30, SIGN: 31,
  SIN: 32, COS: 33, TAN: 34, ASIN: 35, ACOS: 36, ATAN: 37, NOT: 38,
};
const UNARY_OP = { abs: OP.ABS, ceil: OP.CEIL, floor: OP.FLOOR, round: OP.ROUND, trunc: OP.TRUNC, sqrt: OP.SQRT, cbrt: OP.CBRT, exp: OP.EXP, ln: OP.LN, log: OP.LN, log10: OP.LOG10, log2: OP.LOG2, sign: OP.SIGN, sin: OP.SIN, cos: OP.COS, tan: OP.TAN, asin: OP.ASIN, acos: OP.ACOS, atan: OP.ATAN, not: OP.NOT };

const truthy = (v) => v !== 0 && !Number.isNaN(v) && v !== false && v !== undefined && v !== null;
const num = (v) => (v === true ? 1 : v === false ? 0 : Number(v));
const nanAdd = (a, b) => (Number.isNaN(a) && !Number.isNaN(b) ? b : !Number.isNaN(a) && Number.isNaN(b) ? a : a + b); // src/parser.js:18-23
const roundTo = (v, digits) => {
  const f = Math.pow(10, Math.trunc(digits || 0));
  return Math.round(v * f) / f;
};

const UNARY_FN = {
  abs: Math.abs, ceil: Math.ceil, floor: Math.floor, round: Math.round, trunc: Math.trunc, sqrt: Math.sqrt, cbrt: Math.cbrt, exp: Math.exp,
  ln: Math.log, log: Math.log, log10: Math.log10, log2: Math.log2, sign: Math.sign, sin: Math.sin, cos: Math.cos, tan: Math.tan,
  asin: Math.asin, acos: Math.acos, atan: Math.atan, not: (v) => (truthy(v) ? 0 : 1),
};

// ------------------------------------------------------------------ tokens
function tokenize(text) {
  const tokens = [];
  const re = /\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|([A-Za-z_$][A-Za-z0-9_$.]*)|(\|\||[-+*/%^(),?:]))/y;
  let at = 0;
  while (at < text.length) {
    if (/^\s*$/.test(text.slice(at))) break;
    re.lastIndex = at;
    const m = re.exec(text);
    if (!m) throw new Error(`parse error [${at + 1}]: Unknown character "${text.slice(at).trim()[0]}"`);
    if (m[1] !== undefined) tokens.push({ kind: 'num', value: Number(m[1]) });
    else if (m[2] !== undefined) tokens.push({ kind: 'name', value: m[2] });
    else tokens.push({ kind: 'op', value: m[3] });
    at = re.lastIndex;
  }
  tokens.push({ kind: 'end' });
  return tokens;
}

// ------------------------------------------------------------------ parser (recursive descent)
function parseFormula(text) {
  const tokens = tokenize(String(text));
  let pos = 0;
  const peek = () => tokens[pos];
  const isOp = (v) => peek().kind === 'op' && peek().value === v;
  const take = () => tokens[pos++];
  const expectOp = (v) => {
    if (!isOp(v)) throw new Error(`parse error: Expected ${v}`);
    take();
  };

  function conditional() {
    const test = additive();
    if (!isOp('?')) return test;
    take();
    const yes = conditional();
    expectOp(':');
    const no = conditional();
    return { t: 'sel', a: test, b: yes, c: no };
  }
  function additive() {
    let left = term();
    while (isOp('+') || isOp('-') || isOp('||')) {
      const op = take().value;
      left = { t: 'bin', op, a: left, b: term() };
    }
    return left;
  }
  function term() {
    let left = factor();
    while (isOp('*') || isOp('/') || isOp('%')) {
      const op = take().value;
      left = { t: 'bin', op, a: left, b: factor() };
    }
    return left;
  }
  function factor() {
    if (isOp('-') || isOp('+')) {
      const op = take().value;
      const operand = factor();
      return op === '-' ? { t: 'neg', a: operand } : operand;
    }
    const tok = peek();
    if (tok.kind === 'name' && UNARY.includes(tok.value) && !(tokens[pos + 1].kind === 'op' && tokens[pos + 1].value === '(')) {
      take();
      return { t: 'un', fn: tok.value, a: factor() };
    }
    return power();
  }
  function power() {
    const base = atom();
    if (!isOp('^')) return base;
    take();
    return { t: 'bin', op: '^', a: base, b: factor() }; // right-associative, binds tighter than unary minus on its right
  }
  function atom() {
    const tok = take();
    if (tok.kind === 'num') return { t: 'num', v: tok.value };
    if (tok.kind === 'op' && tok.value === '(') {
      const inner = conditional();
      expectOp(')');
      return inner;
    }
    if (tok.kind === 'name') {
      if (isOp('(')) {
        take();
        const args = [];
        if (!isOp(')')) {
          do {
            args.push(conditional());
          } while (isOp(',') && take());
        }
        expectOp(')');
        if (UNARY.includes(tok.value)) {
          if (args.length !== 1) throw new Error(`parse error: ${tok.value} takes one argument`);
          return { t: 'un', fn: tok.value, a: args[0] };
        }
        if (!Object.prototype.hasOwnProperty.call(CALLS, tok.value)) throw new Error(`parse error: unknown function ${tok.value}`);
        const arity = CALLS[tok.value];
        if ((arity >= 0 && args.length !== arity) || (arity < 0 && args.length === 0)) throw new Error(`parse error: wrong number of arguments for ${tok.value}`);
        return { t: 'call', fn: tok.value, args };
      }
      if (Object.prototype.hasOwnProperty.call(CONSTANTS, tok.value)) return { t: 'num', v: CONSTANTS[tok.value], name: tok.value };
      return { t: 'var', name: tok.value };
    }
    throw new Error('parse error: Unexpected ' + (tok.kind === 'end' ? 'end of formula' : String(tok.value)));
  }

  const tree = conditional();
  if (peek().kind !== 'end') throw new Error(`parse error: Unexpected ${peek().value}`);
  return tree;
}

// ------------------------------------------------------------------ tree walkers
function evaluate(node, params) {
  switch (node.t) {
    case 'num':
      return node.v;
    case 'var': {
      if (!(node.name in params)) throw new Error(`undefined variable: ${node.name}`);
      return num(params[node.name]);
    }
    case 'neg':
      return -evaluate(node.a, params);
    case 'un':
      return UNARY_FN[node.fn](evaluate(node.a, params));
    case 'sel':
      return truthy(evaluate(node.a, params)) ? evaluate(node.b, params) : evaluate(node.c, params);
    case 'bin': {
      const a = evaluate(node.a, params);
      const b = evaluate(node.b, params);
      switch (node.op) {
        case '+': return a + b;
        case '-': return a - b;
        case '*': return a * b;
        case '/': return a / b;
        case '%': return a % b;
        case '^': return Math.pow(a, b);
        default: return nanAdd(a, b);
      }
    }
    default: {
      const args = node.args.map((x) => evaluate(x, params));
      switch (node.fn) {
        case 'min': return Math.min(...args);
        case 'max': return Math.max(...args);
        case 'pow': return Math.pow(args[0], args[1]);
        case 'atan2': return Math.atan2(args[0], args[1]);
        case 'hypot': return Math.hypot(...args);
        case 'if': return truthy(args[0]) ? args[1] : args[2];
        case 'roundTo': return roundTo(args[0], args[1]);
        default: return Number.isNaN(args[0]) ? 1 : 0;
      }
    }
  }
}

function show(node) {
  switch (node.t) {
    case 'num':
      return node.name || String(node.v);
    case 'var':
      return node.name;
    case 'neg':
      return `(-${show(node.a)})`;
    case 'un':
      return `(${node.fn} ${show(node.a)})`;
    case 'sel':
      return `(${show(node.a)} ? (${show(node.b)}) : (${show(node.c)}))`;
    case 'bin':
      return `(${show(node.a)} ${node.op} ${show(node.b)})`;
    default:
      return `${node.fn}(${node.args.map(show).join(', ')})`;
  }
}

function collect(node, out) {
  if (node.t === 'var') {
    if (!out.includes(node.name)) out.push(node.name);
  } else {
    for (const key of ['a', 'b', 'c']) if (node[key]) collect(node[key], out);
    if (node.args) node.args.forEach((x) => collect(x, out));
  }
  return out;
}

function replaceVar(node, name, replacement) {
  if (node.t === 'var') return node.name === name ? replacement : node;
  const copy = Object.assign({}, node);
  for (const key of ['a', 'b', 'c']) if (node[key]) copy[key] = replaceVar(node[key], name, replacement);
  if (node.args) copy.args = node.args.map((x) => replaceVar(x, name, replacement));
  return copy;
}

// ------------------------------------------------------------------ lowering to the device program
function lower(node, ctx) {
  const emit = (...words) => ctx.code.push(...words);
  switch (node.t) {
    case 'num': {
      let k = ctx.consts.findIndex((c) => Object.is(c, node.v));
      if (k === -1) k = ctx.consts.push(node.v) - 1;
      emit(OP.CONST, k);
      return;
    }
    case 'var':
      if (node.name in ctx.inputs) emit(OP.INPUT, ctx.inputs[node.name]);
      else if (node.name in ctx.scalars) emit(OP.SCALAR, ctx.scalars[node.name]);
      else throw new Error(`undefined variable: ${node.name}`);
      return;
    case 'neg':
      lower(node.a, ctx);
      emit(OP.NEG);
      return;
    case 'un':
      lower(node.a, ctx);
      emit(UNARY_OP[node.fn]);
      return;
    case 'sel':
      lower(node.a, ctx);
      lower(node.b, ctx);
      lower(node.c, ctx);
      emit(OP.SELECT);
      return;
    case 'bin':
      lower(node.a, ctx);
      lower(node.b, ctx);
      emit({ '+': OP.ADD, '-': OP.SUB, '*': OP.MUL, '/': OP.DIV, '%': OP.MOD, '^': OP.POW, '||': OP.NANADD }[node.op]);
      return;
    default: {
      const fold = { min: OP.MIN, max: OP.MAX, hypot: OP.HYPOT }[node.fn];
      if (fold !== undefined) {
        lower(node.args[0], ctx);
        if (node.fn === 'hypot' && node.args.length === 1) {
          emit(OP.ABS);
          return;
        }
        for (let i = 1; i < node.args.length; ++i) {
          lower(node.args[i], ctx);
          emit(fold); // min/max/hypot fold left: hypot(hypot(a,b),c) == hypot(a,b,c)
        }
        return;
      }
      node.args.forEach((x) => lower(x, ctx));
      emit({ pow: OP.POW, atan2: OP.ATAN2, if: OP.SELECT, roundTo: OP.ROUNDTO, isNaN: OP.ISNAN }[node.fn]);
    }
  }
}

function stackDepth(code) {
  let depth = 0;
  let max = 0;
  for (let i = 0; i < code.length; ++i) {
    const op = code[i];
    if (op <= OP.SCALAR) {
      ++depth;
      ++i; // operand word
    } else if (op === OP.SELECT) depth -= 2;
    else if (op === OP.NEG || op === OP.ISNAN || op >= OP.ABS) depth += 0;
    else depth -= 1; // binary
    if (depth > max) max = depth;
  }
  return max;
}

class Expression {
  constructor(tree) {
    this._tree = tree;
  }

  /** Identifiers that are not functions or constants, in order of first appearance. */
  variables(_options) {
    return collect(this._tree, []);
  }

  evaluate(params = {}) {
    return evaluate(this._tree, params);
  }

  toString() {
    return show(this._tree);
  }

  substitute(variable, replacement) {
    const node = replacement instanceof Expression ? replacement._tree : parseFormula(String(replacement));
    return new Expression(replaceVar(this._tree, variable, node));
  }

  /**
   * Postfix program for the device: `inputs` maps measure ids to input slots, `scalars` maps
   * `<id>__total` names to scalar slots.  Returns { code: Int32Array, consts: Float64Array, depth }.
   */
  compile(inputs, scalars = {}) {
    const ctx = { code: [], consts: [], inputs, scalars };
    lower(this._tree, ctx);
    return { code: Int32Array.from(ctx.code), consts: Float64Array.from(ctx.consts), depth: stackDepth(ctx.code) };
  }
}

class Parser {
  parse(text) {
    return new Expression(parseFormula(text));
  }

  evaluate(text, params) {
    return this.parse(text).evaluate(params);
  }
}

function getParser() {
  return new Parser();
}

module.exports = { getParser, Parser, Expression, OP };
