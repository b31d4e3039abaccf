'use strict';
// src/passes/accumulate.ts -- the running mean.  The textures it owned
// (accumulationTexture / outputTexturePrev, :45-59) live inside libmi3pt.so.
const { Pass } = require('./pass');
const { StructuredView } = require('../layout');

const PASS_ACCUMULATE = 1, SUBMIT_ACCUMULATE = 2;

class AccumulatePass extends Pass {
  constructor(renderer) {
    super(renderer);
    this.passId = PASS_ACCUMULATE;
    this.uniforms = new StructuredView('AccumulateUniforms');
  }
  setUniforms(value) {                       // accumulate.ts:178-188
    this.uniforms.set(value);
    this.renderer.native.setUniforms(this.renderer.handle, this.passId, this.uniforms.bytes);
  }
  update() {                                 // accumulate.ts:190-195
    this.setUniforms({
      resolution: [this.renderer.scaledWidth, this.renderer.scaledHeight],
      frame: this.renderer.frame,
    });
  }
  render(commandEncoder) { commandEncoder.passes |= SUBMIT_ACCUMULATE; }    // accumulate.ts:154-176
}
module.exports = { AccumulatePass };
