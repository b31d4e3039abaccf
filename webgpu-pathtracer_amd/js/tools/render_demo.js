'use strict';
// Headless equivalent of the reference's main loop (src/main.ts:374-400) on the default
// scene: N calls of renderer.render(scene, camera), then read-back.
//   node render_demo.js --env env.f32 --width 64 --height 64 --frames 3 --bounces 4 --out prefix
//   --hdr file.hdr     a Radiance map (1024x512) instead of raw float texels (main.ts:41-46)
//   --model file.glb   replace the default meshes by a loaded model (.glb/.gltf/.obj), main.ts:251-279
//   --devices 0,1,2    a device group: Renderer.create({ devices: [0, 1, 2] }) -- the same loop on several GPUs
// Writes <prefix>.acc.f32 (accumulation, RGBA float), <prefix>.canvas.rgba8 and prints a
// JSON summary.  Needs a HIP device.
const fs = require('fs');
const pt = require('..');
const { buildDefaultScene, PARAMS } = require('../examples/default_scene');

function arg(name, dflt) {
  const i = process.argv.indexOf('--' + name);
  return i >= 0 ? process.argv[i + 1] : dflt;
}

async function main() {
  const width = parseInt(arg('width', '256'), 10), height = parseInt(arg('height', '256'), 10);
  const frames = parseInt(arg('frames', '4'), 10), bounces = parseInt(arg('bounces', String(PARAMS.maxBounces)), 10);
  const envPath = arg('env', null), out = arg('out', 'demo');
  let envData = null;
  if (envPath) {
    const b = fs.readFileSync(envPath);
    envData = new Float32Array(b.buffer, b.byteOffset, b.length / 4);
  }
  const diag = await pt.Renderer.diagnostic();
  if (!diag.supported) throw new Error('HIP device not found.');
  const devices = arg('devices', null);
  const renderer = await pt.Renderer.create(devices ? { enableTimestampQuery: true, devices: devices.split(',').map((d) => parseInt(d, 10)) }
    : { enableTimestampQuery: true });
  const { scene, camera } = buildDefaultScene(envData);
  if (arg('hdr', null)) scene.environment = new pt.RGBELoader().setDataType(pt.FloatType).load(arg('hdr'));
  const modelPath = arg('model', null);
  if (modelPath) {
    const model = /\.obj$/i.test(modelPath) ? new pt.OBJLoader().load(modelPath) : new pt.GLTFLoader().load(modelPath).scene;
    pt.placeModel(model);
    scene.clear();
    scene.add(model);
    scene.needsUpdate = true;
  }
  const events = [];
  for (const ev of ['start', 'reset', 'progress', 'complete', 'resize']) renderer.on(ev, () => events.push(ev));
  renderer.frames = frames;
  renderer.scalingFactor = parseFloat(arg('scale', '1'));
  renderer.setUniforms('raytrace', { maxBounces: bounces, envMapIntensity: PARAMS.envMapIntensity });
  renderer.setUniforms('accumulate', { enabled: PARAMS.accumulate ? 1 : 0 });
  renderer.setUniforms('fullscreen', { denoise: PARAMS.denoise ? 1 : 0, tonemapping: PARAMS.tonemapping });
  renderer.resize(width, height);
  const t0 = Date.now();
  for (let i = 0; i < frames + 1; i++) renderer.render(scene, camera);     // the last call only presents
  const acc = renderer.readAccumulation();
  const ms = Date.now() - t0;
  const canvas = renderer.readCanvas();
  fs.writeFileSync(out + '.acc.f32', Buffer.from(acc.buffer));
  fs.writeFileSync(out + '.canvas.rgba8', Buffer.from(canvas.buffer));
  renderer.screenshot(out + '.png');
  const summary = {
    width, height, frames, status: renderer.status, frame: renderer.frame, events,
    counters: renderer.counters(), stats: renderer.passes.raytrace.stats, wall_ms: ms,
    timings_us: { raytrace: renderer.timings.raytrace.value, accumulate: renderer.timings.accumulate.value,
      fullscreen: renderer.timings.fullscreen.value },
    device: diag.info.description,
  };
  console.log(JSON.stringify(summary));
  await renderer.destroy();
}

main().catch((err) => { console.error(err.stack || String(err)); process.exit(1); });
