'use strict';
// Loads a model (.glb / .gltf / .obj) the way src/main.ts:251-279 does, flattens it like
// RaytracePass.updateScene and writes triangles.bin / materials.bin (+ nodes.bin from the native
// builder); with --hdr <file> also decodes a Radiance .hdr to env.f32.  No GPU needed.
const fs = require('fs');
const path = require('path');
const pt = require('..');

const args = process.argv.slice(2);
let hdr = null;
const rest = [];
for (let i = 0; i < args.length; i++) { if (args[i] === '--hdr') hdr = args[++i]; else rest.push(args[i]); }
const [file, out] = rest;
if (!out) { console.error('usage: node dump_model.js <model|-> <outdir> [--hdr file.hdr]'); process.exit(2); }
const summary = {};
if (file !== '-') {
  const model = /\.obj$/i.test(file) ? new pt.OBJLoader().load(file) : new pt.GLTFLoader().load(file).scene;
  pt.placeModel(model);
  const scene = new pt.RaytracingScene();
  scene.add(model);
  const flat = pt.RaytracePass.flattenScene(scene);
  const packed = pt.RaytracePass.packScene(flat);
  const nodes = pt.loadNative().hostBuildBvhF64(packed.positions, 2);
  fs.writeFileSync(path.join(out, 'triangles.bin'), Buffer.from(packed.triangleBytes));
  fs.writeFileSync(path.join(out, 'materials.bin'), Buffer.from(packed.materialBytes));
  fs.writeFileSync(path.join(out, 'nodes.bin'), nodes);
  Object.assign(summary, { triangles: flat.triangles.length, materials: flat.materials.length, nodes: nodes.length / 48,
    scale: model.scale.x });
}
if (hdr) {
  const tex = new pt.RGBELoader().setDataType(pt.FloatType).load(hdr);
  fs.writeFileSync(path.join(out, 'env.f32'), Buffer.from(tex.image.data.buffer));
  Object.assign(summary, { width: tex.image.width, height: tex.image.height });
}
console.log(JSON.stringify(summary));
