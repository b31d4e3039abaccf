'use strict';
// src/passes/fullscreen.ts -- bilateral de-noise + tone-map to the canvas; the canvas here
// is an RGBA8 image inside libmi3pt.so (Renderer.readCanvas()).
const { Pass } = require('./pass');
const { StructuredView } = require('../layout');

const PASS_FULLSCREEN = 2, SUBMIT_FULLSCREEN = 4;

class FullscreenPass extends Pass {
  constructor(renderer) {
    super(renderer);
    this.passId = PASS_FULLSCREEN;
    this.uniforms = new StructuredView('FullscreenUniforms');
  }
  setUniforms(value) {                       // fullscreen.ts:138-148
    this.uniforms.set(value);
    this.renderer.native.setUniforms(this.renderer.handle, this.passId, this.uniforms.bytes);
  }
  update() {                                 // fullscreen.ts:150-156
    this.setUniforms({
      resolution: [this.renderer.width, this.renderer.height],
      aspect: this.renderer.aspect,
      scalingFactor: this.renderer.scalingFactor,
    });
  }
  render(commandEncoder) { commandEncoder.passes |= SUBMIT_FULLSCREEN; }    // fullscreen.ts:158-177
}
module.exports = { FullscreenPass };
