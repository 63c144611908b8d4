'use strict';
/*
 * calendar.js — time slots for TimeDimension.
 *
 * The reference delegates all calendar arithmetic to the third-party module `timeslot-dag`
 * (pinned 2.2.0 in /root/reference/package-lock.json:7284-7291; call sites
 * src/dimension/time.js:7,19-20,55-61,71,92,118-132,150-166,189-192).  That module is not
 * vendored and is absent offline, so this is a restatement of its published behaviour:
 *
 *   day            2010-01-31          week_sat|sun|mon    2010-W05-mon  (epidemiological weeks:
 *   month          2010-01                                 week 1 is the week holding January 4th)
 *   quarter        2010-Q1             month_week_*        2010-01-W1-mon (weeks cut at month ends)
 *   semester       2010-S1             year                2010           all   all
 *
 * PARITY: pinned only by the literals of the reference's tests (test/dimension-time.js:26-88,
 * 157-221, test/cube-drilling.js:26-189, test/cube-dimension.js:58-63); everything else here is
 * plain Gregorian arithmetic and is otherwise "parity unpinned" (SURVEY.md §8(c)).
 */

const DAY_MS = 86400000;
const WEEK_START = { sat: 6, sun: 0, mon: 1 }; // Date#getUTCDay numbering

// children -> parents (the DAG): a slot of periodicity k can be rolled up to any of PARENTS[k]
const PARENTS = {
  day: ['week_sat', 'week_sun', 'week_mon', 'month_week_sat', 'month_week_sun', 'month_week_mon', 'month', 'quarter', 'semester', 'year', 'all'],
  week_sat: ['month', 'quarter', 'semester', 'year', 'all'],
  week_sun: ['month', 'quarter', 'semester', 'year', 'all'],
  week_mon: ['month', 'quarter', 'semester', 'year', 'all'],
  month_week_sat: ['week_sat', 'month', 'quarter', 'semester', 'year', 'all'],
  month_week_sun: ['week_sun', 'month', 'quarter', 'semester', 'year', 'all'],
  month_week_mon: ['week_mon', 'month', 'quarter', 'semester', 'year', 'all'],
  month: ['quarter', 'semester', 'year', 'all'],
  quarter: ['semester', 'year', 'all'],
  semester: ['year', 'all'],
  year: ['all'],
  all: [],
};

const PATTERNS = [
  ['day', /^(\d{4})-(\d{2})-(\d{2})$/],
  ['month_week', /^(\d{4})-(\d{2})-W(\d)-(sat|sun|mon)$/],
  ['week', /^(\d{4})-W(\d{2})-(sat|sun|mon)$/],
  ['month', /^(\d{4})-(\d{2})$/],
  ['quarter', /^(\d{4})-Q([1-4])$/],
  ['semester', /^(\d{4})-S([12])$/],
  ['year', /^(\d{4})$/],
  ['all', /^all$/],
];

const utc = (y, m, d) => Date.UTC(y, m, d); // m is 0-based, d may overflow (normalised by Date)
const pad2 = (n) => (n < 10 ? '0' + n : String(n));
const pad4 = (n) => ('0000' + n).slice(-4);

function ymd(ms) {
  const d = new Date(ms);
  return [d.getUTCFullYear(), d.getUTCMonth(), d.getUTCDate(), d.getUTCDay()];
}

// first day of week 1 of `year` for weeks starting on weekday `dow`: the week that holds Jan 4th
function weekEpoch(year, dow) {
  const jan4 = utc(year, 0, 4);
  const back = (new Date(jan4).getUTCDay() - dow + 7) % 7;
  return jan4 - back * DAY_MS;
}

// length of the (possibly partial) first week of a month for weeks starting on `dow`
function firstMonthWeekLength(year, month, dow) {
  const first = new Date(utc(year, month, 1)).getUTCDay();
  return ((dow - first + 7) % 7) || 7;
}

class TimeSlot {
  constructor(value, periodicity, firstMs, lastMs) {
    this.value = value;
    this.periodicity = periodicity;
    this._first = firstMs;
    this._last = lastMs;
  }

  get firstDate() {
    return new Date(this._first);
  }

  get lastDate() {
    return new Date(this._last);
  }

  static get upperSlots() {
    return PARENTS;
  }

  /** Parses a slot value; throws on anything that is not one of the formats above. */
  static fromValue(value) {
    const text = String(value);
    for (const [kind, re] of PATTERNS) {
      const m = re.exec(text);
      if (!m) continue;
      const y = Number(m[1]);
      switch (kind) {
        case 'day': {
          const t = utc(y, Number(m[2]) - 1, Number(m[3]));
          return new TimeSlot(text, 'day', t, t);
        }
        case 'week': {
          const first = weekEpoch(y, WEEK_START[m[3]]) + (Number(m[2]) - 1) * 7 * DAY_MS;
          return new TimeSlot(text, 'week_' + m[3], first, first + 6 * DAY_MS);
        }
        case 'month_week': {
          const month = Number(m[2]) - 1;
          const n = Number(m[3]);
          const len1 = firstMonthWeekLength(y, month, WEEK_START[m[4]]);
          const monthEnd = utc(y, month + 1, 0);
          const first = n === 1 ? utc(y, month, 1) : utc(y, month, 1 + len1 + (n - 2) * 7);
          const last = Math.min(n === 1 ? utc(y, month, len1) : first + 6 * DAY_MS, monthEnd);
          return new TimeSlot(text, 'month_week_' + m[4], first, last);
        }
        case 'month':
          return new TimeSlot(text, 'month', utc(y, Number(m[2]) - 1, 1), utc(y, Number(m[2]), 0));
        case 'quarter': {
          const q = Number(m[2]) - 1;
          return new TimeSlot(text, 'quarter', utc(y, q * 3, 1), utc(y, q * 3 + 3, 0));
        }
        case 'semester': {
          const s = Number(m[2]) - 1;
          return new TimeSlot(text, 'semester', utc(y, s * 6, 1), utc(y, s * 6 + 6, 0));
        }
        case 'year':
          return new TimeSlot(text, 'year', utc(y, 0, 1), utc(y, 11, 31));
        default:
          return new TimeSlot('all', 'all', utc(0, 0, 1), utc(9999, 11, 31));
      }
    }
    throw new Error(`Invalid time slot value: ${value}`);
  }

  /** The slot of the given periodicity that contains the (UTC) date. */
  static fromDate(date, periodicity) {
    const ms = date instanceof Date ? date.getTime() : Number(date);
    const day = Math.floor(ms / DAY_MS) * DAY_MS;
    const [y, m, d] = ymd(day);
    switch (periodicity) {
      case 'day':
        return TimeSlot.fromValue(`${pad4(y)}-${pad2(m + 1)}-${pad2(d)}`);
      case 'week_sat':
      case 'week_sun':
      case 'week_mon': {
        const suffix = periodicity.slice(-3);
        let year = y + 1;
        let epoch = weekEpoch(year, WEEK_START[suffix]);
        while (day < epoch) epoch = weekEpoch(--year, WEEK_START[suffix]);
        const week = Math.floor((day - epoch) / (7 * DAY_MS)) + 1;
        return TimeSlot.fromValue(`${pad4(year)}-W${pad2(week)}-${suffix}`);
      }
      case 'month_week_sat':
      case 'month_week_sun':
      case 'month_week_mon': {
        const suffix = periodicity.slice(-3);
        const len1 = firstMonthWeekLength(y, m, WEEK_START[suffix]);
        const n = d <= len1 ? 1 : Math.floor((d - 1 - len1) / 7) + 2;
        return TimeSlot.fromValue(`${pad4(y)}-${pad2(m + 1)}-W${n}-${suffix}`);
      }
      case 'month':
        return TimeSlot.fromValue(`${pad4(y)}-${pad2(m + 1)}`);
      case 'quarter':
        return TimeSlot.fromValue(`${pad4(y)}-Q${Math.floor(m / 3) + 1}`);
      case 'semester':
        return TimeSlot.fromValue(`${pad4(y)}-S${Math.floor(m / 6) + 1}`);
      case 'year':
        return TimeSlot.fromValue(pad4(y));
      case 'all':
        return TimeSlot.fromValue('all');
      default:
        throw new Error(`Invalid periodicity: ${periodicity}`);
    }
  }

  /**
   * Roll this slot up.  Full weeks belong to the month/quarter/year of their middle day
   * (first day + 3), every other slot to that of its first day.
   */
  toParentPeriodicity(periodicity) {
    if (periodicity === this.periodicity) return this;
    if (PARENTS[this.periodicity].indexOf(periodicity) === -1)
      throw new Error(`Cannot convert ${this.periodicity} to ${periodicity}`);
    const fullWeek = this.periodicity === 'week_sat' || this.periodicity === 'week_sun' || this.periodicity === 'week_mon';
    return TimeSlot.fromDate(this._first + (fullWeek ? 3 * DAY_MS : 0), periodicity);
  }

  next() {
    if (this.periodicity === 'all') throw new Error('There is no slot after "all"');
    return TimeSlot.fromDate(this._last + DAY_MS, this.periodicity);
  }

  previous() {
    if (this.periodicity === 'all') throw new Error('There is no slot before "all"');
    return TimeSlot.fromDate(this._first - DAY_MS, this.periodicity);
  }

  humanizeValue(language = 'en') {
    return humanize(this, language);
  }
}

const MONTHS = {
  en: ['January', 'February', 'March', 'April', 'May', 'June', 'July', 'August', 'September', 'October', 'November', 'December'],
  fr: ['Janvier', 'Février', 'Mars', 'Avril', 'Mai', 'Juin', 'Juillet', 'Août', 'Septembre', 'Octobre', 'Novembre', 'Décembre'],
  es: ['Enero', 'Febrero', 'Marzo', 'Abril', 'Mayo', 'Junio', 'Julio', 'Agosto', 'Septiembre', 'Octubre', 'Noviembre', 'Diciembre'],
};

function humanize(slot, language) {
  const lang = MONTHS[language] ? language : 'en';
  const [y, m] = ymd(slot._first);
  switch (slot.periodicity) {
    case 'all':
      return { en: 'All', fr: 'Tout', es: 'Todo' }[lang];
    case 'year':
      return String(y);
    case 'semester': {
      const n = slot.value.slice(-1);
      if (lang === 'fr') return `${n === '1' ? '1er' : '2ème'} sem. ${y}`;
      if (lang === 'es') return `${n === '1' ? '1er' : '2do'} sem. ${y}`;
      return `${n === '1' ? '1st' : '2nd'} sem. ${y}`;
    }
    case 'quarter': {
      const n = Number(slot.value.slice(-1));
      if (lang === 'fr') return `${n === 1 ? '1er' : n + 'ème'} trim. ${y}`;
      if (lang === 'es') return `${['1er', '2do', '3er', '4to'][n - 1]} trim. ${y}`;
      return `${['1st', '2nd', '3rd', '4th'][n - 1]} qu. ${y}`;
    }
    case 'month':
      return `${MONTHS[lang][m]} ${y}`;
    case 'day':
      return slot.value;
    default: {
      // weeks: "Week 5 2010" style, partial month weeks carry their month
      const mw = /^(\d{4})-(\d{2})-W(\d)-/.exec(slot.value);
      const word = { en: 'Week', fr: 'Sem.', es: 'Sem.' }[lang];
      if (mw) return `${word} ${mw[3]} ${MONTHS[lang][Number(mw[2]) - 1]} ${mw[1]}`;
      const w = /^(\d{4})-W(\d{2})-/.exec(slot.value);
      return `${word} ${Number(w[2])} ${w[1]}`;
    }
  }
}

module.exports = TimeSlot;
