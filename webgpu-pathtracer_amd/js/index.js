'use strict';
// Node host of the MI355X path tracer: the reference's Renderer / Pass / Scene API
// (umar-ahmed/webgpu-pathtracer, src/renderer.ts, src/passes/*.ts, src/scene.ts) over
// libmi3pt.so.  See index.d.ts for the typed surface.
module.exports = Object.assign({},
  require('./src/renderer'),
  require('./src/scene'),
  require('./src/loaders'),
  require('./src/math3'),
  require('./src/layout'),
  require('./src/timing'),
  { Pass: require('./src/passes/pass').Pass },
  require('./src/passes/raytrace'),
  require('./src/passes/accumulate'),
  require('./src/passes/fullscreen'));
