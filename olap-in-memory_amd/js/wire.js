'use strict';
/*
 * wire.js — encoder / decoder of the reference's tagged binary container
 * (/root/reference/src/serialization.js:15-140), so that cubes serialised by either
 * implementation can be read by the other.  The format, all little-endian 32-bit words:
 *
 *   1 ARRAY_BUFFER  [1][byteLength][bytes, zero-padded to a multiple of 4]
 *   2 TYPED_ARRAY   [2][type index in the list below][ARRAY_BUFFER record of its buffer]
 *   3 ARRAY         [3][count] then per item [item byte length][item record]
 *   4 STRING        [4][TYPED_ARRAY record of the UTF-8 bytes (Uint8Array)]
 *   5 OBJECT        [5][ARRAY record of [key, ARRAY_BUFFER holding the value's own record] pairs]
 *   6 NULL          [6]
 *   7 NUMBER        [7][float32]           (numbers lose precision: they are stored as Float32)
 *   8 BOOLEAN       [8][float32 1 or 0]
 *   undefined is written as one zero word and decodes to undefined.
 *
 * Independent implementation (one growing byte sink + DataView) of the published format.
 */
const KINDS = [Int8Array, Uint8Array, Uint8ClampedArray, Int16Array, Uint16Array, Int32Array, Uint32Array, Float32Array, Float64Array,
  typeof BigInt64Array === 'undefined' ? null : BigInt64Array, typeof BigUint64Array === 'undefined' ? null : BigUint64Array];

class Sink {
  constructor() {
    this.bytes = new Uint8Array(256);
    this.view = new DataView(this.bytes.buffer);
    this.length = 0;
  }

  reserve(extra) {
    if (this.length + extra <= this.bytes.length) return;
    let cap = this.bytes.length * 2;
    while (cap < this.length + extra) cap *= 2;
    const grown = new Uint8Array(cap);
    grown.set(this.bytes.subarray(0, this.length));
    this.bytes = grown;
    this.view = new DataView(grown.buffer);
  }

  u32(value) {
    this.reserve(4);
    this.view.setUint32(this.length, value >>> 0, true);
    this.length += 4;
  }

  f32(value) {
    this.reserve(4);
    this.view.setFloat32(this.length, value, true);
    this.length += 4;
  }

  raw(u8, padTo4) {
    const padded = padTo4 ? Math.ceil(u8.length / 4) * 4 : u8.length;
    this.reserve(padded);
    this.bytes.set(u8, this.length);
    this.bytes.fill(0, this.length + u8.length, this.length + padded);
    this.length += padded;
  }

  /** Reserves a 32-bit length slot, runs `body`, then back-patches the number of bytes it wrote. */
  sized(body) {
    const slot = this.length;
    this.u32(0);
    body();
    this.view.setUint32(slot, this.length - slot - 4, true);
  }

  done() {
    return this.bytes.buffer.slice(0, this.length);
  }
}

function write(sink, value) {
  if (value === null) {
    sink.u32(6);
  } else if (value === undefined) {
    sink.u32(0);
  } else if (value instanceof ArrayBuffer) {
    sink.u32(1);
    sink.u32(value.byteLength);
    sink.raw(new Uint8Array(value), true);
  } else if (ArrayBuffer.isView(value)) {
    sink.u32(2);
    sink.u32(KINDS.findIndex((K) => K && value instanceof K));
    // the reference serialises the WHOLE underlying buffer (obj.buffer); copy views that are windows
    const whole = value.byteOffset === 0 && value.byteLength === value.buffer.byteLength ? value.buffer : value.buffer.slice(value.byteOffset, value.byteOffset + value.byteLength);
    write(sink, whole);
  } else if (Array.isArray(value)) {
    sink.u32(3);
    sink.u32(value.length);
    for (const item of value) sink.sized(() => write(sink, item));
  } else if (typeof value === 'string') {
    sink.u32(4);
    write(sink, Uint8Array.from(Buffer.from(value, 'utf8')));
  } else if (typeof value === 'number') {
    sink.u32(7);
    sink.f32(value);
  } else if (typeof value === 'boolean') {
    sink.u32(8);
    sink.f32(value ? 1 : 0);
  } else {
    sink.u32(5);
    write(sink, Object.entries(value).map(([key, item]) => [key, toBuffer(item)]));
  }
}

function toBuffer(value) {
  const sink = new Sink();
  write(sink, value);
  return sink.done();
}

function read(view, at) {
  const tag = view.getUint32(at, true);
  switch (tag) {
    case 1: {
      const n = view.getUint32(at + 4, true);
      const start = view.byteOffset + at + 8;
      return view.buffer.slice(start, start + n);
    }
    case 2:
      return new KINDS[view.getUint32(at + 4, true)](read(view, at + 8));
    case 3: {
      const count = view.getUint32(at + 4, true);
      const out = [];
      let cursor = at + 8;
      for (let i = 0; i < count; ++i) {
        const size = view.getUint32(cursor, true);
        out.push(read(view, cursor + 4));
        cursor += 4 + size;
      }
      return out;
    }
    case 4:
      return Buffer.from(read(view, at + 4)).toString('utf8');
    case 5: {
      const out = {};
      for (const [key, blob] of read(view, at + 4)) out[key] = fromBuffer(blob);
      return out;
    }
    case 6:
      return null;
    case 7:
      return view.getFloat32(at + 4, true);
    case 8:
      return view.getFloat32(at + 4, true) === 1;
    default:
      return undefined;
  }
}

function fromBuffer(buffer, offset = 0) {
  const ab = buffer instanceof ArrayBuffer ? buffer : buffer.buffer.slice(buffer.byteOffset, buffer.byteOffset + buffer.byteLength);
  return read(new DataView(ab), offset);
}

/** Node Buffer -> ArrayBuffer (the reference's helper of the same name). */
function toArrayBuffer(buf) {
  return buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
}

module.exports = { toBuffer, fromBuffer, toArrayBuffer };
