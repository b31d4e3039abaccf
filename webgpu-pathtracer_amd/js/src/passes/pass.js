'use strict';
// src/passes/pass.ts:4-27 -- the Pass contract: render(commandEncoder) encodes only,
// update() pushes per-frame uniforms, updateTimings() feeds the rolling average (µs).
const { RollingAverage } = require('../timing');

class Pass {
  constructor(renderer) {
    this.renderer = renderer;
    this.timingAverage = new RollingAverage();
  }
  render(/* commandEncoder */) { throw new Error('abstract'); }
  update() { throw new Error('abstract'); }
  updateTimings() {
    if (!this.renderer.options.enableTimestampQuery) return;
    const us = this.renderer.native.passTimeUs(this.renderer.handle, this.passId);
    if (us !== null) this.timingAverage.addSample(us);   // null: fused into another kernel this frame
  }
}
module.exports = { Pass };
