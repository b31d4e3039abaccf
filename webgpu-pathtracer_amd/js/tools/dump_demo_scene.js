'use strict';
// Writes the flattened default scene exactly as RaytracePass.updateScene would upload it
// (no GPU needed): triangles.bin (112 B), materials.bin (64 B), nodes.bin (48 B, native
// builder), camera.bin (the 96-B raytrace uniform block with only the camera fields set).
const fs = require('fs');
const path = require('path');
const pt = require('..');
const { buildDefaultScene } = require('../examples/default_scene');

const out = process.argv[2];
if (!out) { console.error('usage: node dump_demo_scene.js <outdir>'); process.exit(2); }
const { scene, camera } = buildDefaultScene(null);
const flat = pt.RaytracePass.flattenScene(scene);
const packed = pt.RaytracePass.packScene(flat);
const native = pt.loadNative();
const nodes = native.hostBuildBvhF64(packed.positions, 2);
const u = new pt.StructuredView('RaytraceUniforms');
u.set({ camera: {
  position: camera.getWorldPosition(new pt.Vector3()).toArray(),
  direction: camera.getWorldDirection(new pt.Vector3()).toArray(),
  fov: camera.fov, focalDistance: camera.focalDistance, aperture: camera.aperture } });
fs.writeFileSync(path.join(out, 'triangles.bin'), Buffer.from(packed.triangleBytes));
fs.writeFileSync(path.join(out, 'materials.bin'), Buffer.from(packed.materialBytes));
fs.writeFileSync(path.join(out, 'nodes.bin'), nodes);
fs.writeFileSync(path.join(out, 'camera.bin'), Buffer.from(u.bytes));
console.log(JSON.stringify({ triangles: flat.triangles.length, materials: flat.materials.length, nodes: nodes.length / 48 }));
