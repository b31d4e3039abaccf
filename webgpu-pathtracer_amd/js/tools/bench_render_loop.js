'use strict';
// Times the drop-in loop exactly as the reference drives it (src/renderer.ts:366-395 called once
// per animation frame, src/main.ts:387-400): renderer.frames sample frames + one more render()
// after sampling has stopped, on the default scene.  Four legs:
//   present_latest   presentEveryFrame true, presentLatest true  (the headless default: the canvas is
//                    drawn once per launched batch; the last, idle render() shows every frame)
//   present_exact    presentEveryFrame true, presentLatest false (the reference's canvas semantics:
//                    one raytrace launch + one fullscreen pass per render())
//   no_present       presentEveryFrame false (the fullscreen pass is never encoded)
//   c_abi            the same frames pushed through the addon directly (setUniforms + submit), no
//                    Renderer / Pass objects: what bench.py measures from Python
// and checks that present_latest and present_exact end with the same canvas, byte for byte.
//   node bench_render_loop.js --env env.f32 [--width 1920 --height 1080 --frames 64 --bounces 8]
const fs = require('fs');
const pt = require('..');
const { buildDefaultScene, PARAMS } = require('../examples/default_scene');

function arg(name, dflt) {
  const i = process.argv.indexOf('--' + name);
  return i >= 0 ? process.argv[i + 1] : dflt;
}

async function leg(name, options, width, height, frames, bounces, envData) {
  const renderer = await pt.Renderer.create(options);
  const { scene, camera } = buildDefaultScene(envData);
  renderer.frames = frames;
  renderer.scalingFactor = 1;
  renderer.setUniforms('raytrace', { maxBounces: bounces, envMapIntensity: PARAMS.envMapIntensity });
  renderer.setUniforms('accumulate', { enabled: 1 });
  renderer.setUniforms('fullscreen', { denoise: 1, tonemapping: PARAMS.tonemapping });
  renderer.resize(width, height);
  for (let i = 0; i < frames + 1; i++) renderer.render(scene, camera);        // warm-up pass of the whole loop
  renderer.sync();
  renderer.reset();
  renderer.native.resetCounters(renderer.handle);
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < frames + 1; i++) renderer.render(scene, camera);        // the last call only presents
  renderer.sync();
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  const rays = Number(renderer.counters().rays);
  const canvas = options.presentEveryFrame === false ? null : Buffer.from(renderer.readCanvas());
  const acc = Buffer.from(renderer.readAccumulation().buffer);
  await renderer.destroy();
  return { name, ms_per_frame: ms / frames, mrays_per_s: rays / ms / 1e3, rays, canvas, acc };
}

async function cAbiLeg(width, height, frames, bounces, envData) {
  // the buffers the Renderer would upload, taken from a Renderer that is then only used as a handle
  const renderer = await pt.Renderer.create({ presentEveryFrame: false });
  const { scene, camera } = buildDefaultScene(envData);
  renderer.frames = frames;
  renderer.scalingFactor = 1;
  renderer.setUniforms('raytrace', { maxBounces: bounces, envMapIntensity: PARAMS.envMapIntensity });
  renderer.setUniforms('accumulate', { enabled: 1 });
  renderer.resize(width, height);
  renderer.update(scene, camera);                       // scene compile + upload, camera uniforms
  const native = renderer.native, h = renderer.handle;
  const rt = renderer.passes.raytrace, acc = renderer.passes.accumulate;
  const run = (first) => {
    for (let f = first; f < first + frames; f++) {
      rt.uniforms.set({ resolution: [width, height], aspect: width / height, frame: f, samplesPerFrame: 1 });
      native.setUniforms(h, 0, rt.uniforms.bytes);
      acc.uniforms.set({ resolution: [width, height], frame: f });
      native.setUniforms(h, 1, acc.uniforms.bytes);
      native.submit(h, 3);
    }
    native.sync(h);
  };
  run(2);
  native.reset(h);
  native.resetCounters(h);
  const t0 = process.hrtime.bigint();
  run(2);
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  const rays = Number(renderer.counters().rays);
  const accImg = Buffer.from(renderer.readAccumulation().buffer);
  await renderer.destroy();
  return { name: 'c_abi', ms_per_frame: ms / frames, mrays_per_s: rays / ms / 1e3, rays, canvas: null, acc: accImg };
}

async function main() {
  const width = parseInt(arg('width', '1920'), 10), height = parseInt(arg('height', '1080'), 10);
  const frames = parseInt(arg('frames', '64'), 10), bounces = parseInt(arg('bounces', '8'), 10);
  let envData = null;
  if (arg('env', null)) {
    const b = fs.readFileSync(arg('env'));
    envData = new Float32Array(b.buffer, b.byteOffset, b.length / 4);
  }
  const legs = [
    await leg('present_latest', { presentEveryFrame: true, presentLatest: true }, width, height, frames, bounces, envData),
    await leg('present_exact', { presentEveryFrame: true, presentLatest: false }, width, height, frames, bounces, envData),
    await leg('no_present', { presentEveryFrame: false }, width, height, frames, bounces, envData),
    await cAbiLeg(width, height, frames, bounces, envData),
  ];
  const out = { width, height, frames, bounces, legs: {} };
  for (const l of legs) out.legs[l.name] = { ms_per_frame: +l.ms_per_frame.toFixed(4), mrays_per_s: +l.mrays_per_s.toFixed(1), rays: l.rays };
  out.same_canvas = legs[0].canvas.equals(legs[1].canvas);
  out.same_accumulation = legs.every((l) => l.acc.equals(legs[0].acc));
  console.log(JSON.stringify(out));
}

main().catch((err) => { console.error(err.stack || String(err)); process.exit(1); });
