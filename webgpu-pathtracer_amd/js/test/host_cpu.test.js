'use strict';
// CPU-only checks of the Node host (run by tests/test_node_host.py): layouts, partial
// uniform sets, the frame-counter state machine over a recording fake of the addon,
// validation errors, and the loud failure without a HIP device.
const assert = require('assert');
const pt = require('..');

// --- layouts (SURVEY.md 8a)
assert.strictEqual(pt.STRUCTS.Triangle.size, 112);
assert.strictEqual(pt.STRUCTS.BVHNode.size, 48);
assert.strictEqual(pt.STRUCTS.Material.size, 64);
assert.strictEqual(pt.STRUCTS.RaytraceUniforms.size, 96);
assert.strictEqual(pt.STRUCTS.AccumulateUniforms.size, 16);
assert.strictEqual(pt.STRUCTS.FullscreenUniforms.size, 24);
const u = new pt.StructuredView('RaytraceUniforms');
u.set({ maxBounces: 4, envMapIntensity: 1.5 });
u.set({ camera: { fov: 45, position: [0, 1, 4] }, frame: 7, unknownKey: 1 });
const dv = new DataView(u.arrayBuffer);
assert.strictEqual(dv.getInt32(16, true), 4);           // survived the second partial set
assert.strictEqual(dv.getFloat32(80, true), 1.5);
assert.strictEqual(dv.getUint32(12, true), 7);
assert.strictEqual(dv.getFloat32(60, true), 45);
assert.strictEqual(dv.getFloat32(36, true), 1);
const a = new pt.StructuredView('AccumulateUniforms');
a.set({ resolution: [479.75, 269.99], frame: 3 });
assert.strictEqual(new DataView(a.arrayBuffer).getUint32(0, true), 479);   // ToUint32 truncation
assert.strictEqual(new DataView(a.arrayBuffer).getUint32(4, true), 269);

// --- geometry counts (src/main.ts:61-73)
assert.strictEqual(new pt.PlaneGeometry(5, 5).getIndex().count, 6);
assert.strictEqual(new pt.BoxGeometry(0.8, 0.8, 0.8).getIndex().count, 36);
assert.strictEqual(new pt.SphereGeometry(0.5, 32, 32).getIndex().count, 1984 * 3);

// --- state machine over a fake addon
function fakeNative(log) {
  const handler = { get: (t, name) => (...args) => { log.push([name].concat(args)); if (name === 'tileLocalRows') return 8; if (name === 'passTimeUs') return null; if (name === 'hostBuildBvhF64') return Buffer.alloc(48 * (2 * args[0].length / 9 - 1)); if (name === 'hostEnvCdf') return new Float32Array(1024 * 512 * 4); return undefined; } };
  return new Proxy({}, handler);
}
const log = [];
const r = new pt.Renderer({ native: fakeNative(log), handle: {}, options: {} });
const { buildDefaultScene } = require('../examples/default_scene');
const { scene, camera } = buildDefaultScene(new Float32Array(1024 * 512 * 4));
const events = [];
for (const ev of ['start', 'reset', 'progress', 'complete', 'resize']) r.on(ev, () => events.push(ev));
r.frames = 3;
r.scalingFactor = 1;
assert.strictEqual(r.status, 'idle');
r.resize(64, 32);
assert.strictEqual(r.status, 'sampling');
assert.deepStrictEqual(events.slice(0, 3), ['reset', 'start', 'resize']);
const frames = [];
for (let i = 0; i < 5; i++) {
  r.render(scene, camera);
  const rt = log.filter((c) => c[0] === 'setUniforms' && c[2] === 0).pop()[3];
  frames.push(new DataView(rt.buffer, rt.byteOffset).getUint32(12, true));
}
assert.deepStrictEqual(frames, [2, 3, 4, 4, 4]);          // first sampled frame carries frame = 2
assert.deepStrictEqual(log.filter((c) => c[0] === 'submit').map((c) => c[2]), [7, 7, 7, 4, 4]);
assert.strictEqual(r.status, 'idle');
assert.strictEqual(events.filter((e) => e === 'complete').length, 1);
assert.strictEqual(events.filter((e) => e === 'progress').length, 3);
assert.strictEqual(scene.needsUpdate, false);
assert.deepStrictEqual(log.filter((c) => /^upload/.test(c[0])).map((c) => c[0]),
  ['uploadEnvironment', 'uploadEnvironmentCdf', 'uploadBvh', 'uploadTriangles', 'uploadMaterials']);
assert.deepStrictEqual(r.passes.raytrace.stats, { Triangles: 1998, Materials: 2, 'BVH Nodes': 3995 });
r.reset();
assert.strictEqual(r.frame, 1);
r.pause();
r.render(scene, camera);
assert.strictEqual(r.frame, 1);
r.start();
r.render(scene, camera);
assert.strictEqual(r.frame, 2);
assert.ok(Math.abs(r.progress - 2 / 4) < 1e-12);

// scalingFactor < 1: fractional float resolution for raytrace, truncated u32 for accumulate
r.resize(90, 50);
r.scalingFactor = 0.25;
r.render(scene, camera);
const rt = log.filter((c) => c[0] === 'setUniforms' && c[2] === 0).pop()[3];
assert.strictEqual(new DataView(rt.buffer, rt.byteOffset).getFloat32(0, true), 22.5);
const ac = log.filter((c) => c[0] === 'setUniforms' && c[2] === 1).pop()[3];
assert.strictEqual(new DataView(ac.buffer, ac.byteOffset).getUint32(0, true), 22);

// --- validation (renderer.ts:133-143)
assert.throws(() => r.updateEnvironmentTexture(new pt.DataTexture(new Float32Array(4), 2, 2)), /1024x512/);
assert.throws(() => r.updateEnvironmentTexture(new pt.DataTexture(new Float32Array(4), 1024, 512, 1009)), /floating point/);
const empty = new pt.RaytracingScene();
empty.needsUpdate = true;
assert.throws(() => r.update(empty, camera), /Input nodes array is empty/);

// --- the real addon: loads, fails loudly without a device, host-side builder works
const native = pt.loadNative();
assert.strictEqual(native.abiVersion(), 4);
assert.strictEqual(native.tileLocalRows(70, 1, 3, 5), 25);
// the deal goes back and forth: rank 0 of 3 owns block 0 of round 0 and block 2 of round 1 (rows 25-29 with 5-row blocks)
assert.strictEqual(native.tileGlobalRow(0, 0, 3, 5), 0);
assert.strictEqual(native.tileGlobalRow(5, 0, 3, 5), 25);
assert.strictEqual(native.tileGlobalRow(7, 2, 3, 5), 17);
for (let y = 0; y < 70; y++) {
  const owner = native.tileOwner(y, 3, 5);
  let found = false;
  for (let ly = 0; ly < native.tileLocalRows(70, owner, 3, 5); ly++) if (native.tileGlobalRow(ly, owner, 3, 5) === y) found = true;
  assert.ok(found, `row ${y}`);
}
assert.throws(() => native.hostBuildBvhF64(new Float64Array(0)), /Input nodes array is empty/);
const nodes = native.hostBuildBvhF64(new Float64Array([0, 0, 0, 1, 0, 0, 0, 1, 0, 5, 5, 5, 6, 5, 5, 5, 6, 5]), 1);
assert.strictEqual(nodes.length, 3 * 48);
(async () => {
  if (native.deviceCount() === 0) {
    assert.deepStrictEqual(await pt.Renderer.diagnostic(), { supported: false });
    await assert.rejects(pt.Renderer.create(), /HIP device not found/);
  }
  console.log('host_cpu.test.js ok');
})().catch((e) => { console.error(e); process.exit(1); });
