'use strict';
// src/timing.ts:1-20 -- RollingAverage over the last 30 samples.  The TimingHelper /
// timestamp-query machinery (timing.ts:28-168) has no headless meaning; per-pass GPU time
// comes from HIP events inside libmi3pt.so (mi3pt_pass_time_us).
class RollingAverage {
  constructor(numSamples) {
    this.numSamples = numSamples === undefined ? 30 : numSamples;
    this.samples = [];
    this.cursor = 0;
    this.total = 0;
  }
  addSample(v) {
    this.total += v - (this.samples[this.cursor] || 0);
    this.samples[this.cursor] = v;
    this.cursor = (this.cursor + 1) % this.numSamples;
  }
  get value() { return this.total / this.samples.length; }      // NaN before the first sample, like the reference
}
module.exports = { RollingAverage };
