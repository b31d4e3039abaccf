'use strict';
// TEST INFRASTRUCTURE.  Executes the reference's own TypeScript host code under Node:
//   RaytracePass.buildBVH / buildBVHRecursive / flattenBVH   (src/passes/raytrace.ts)
//   Renderer.updateEnvironmentTexture                        (src/renderer.ts)
// The method texts are cut out of the reference checkout at run time (never copied into this
// repository), their TypeScript annotations are stripped, and they run against a minimal THREE
// shim (Box3 / Vector3 as published in three@0.171.0) and a recording GPUQueue stub.
//
//   node run_reference_host.js bvh <reference-root> <positions.f64> <out-nodes.bin>
//   node run_reference_host.js cdf <reference-root> <env.f32> <out-cdf.f32>
const fs = require('fs');
const path = require('path');

// ------------------------------------------------------------------ cutting and stripping

function extractMethod(source, name) {
  const re = new RegExp('^[ \\t]*(?:private |public |protected )?' + name + '\\s*\\(', 'm');
  const m = re.exec(source);
  if (!m) throw new Error('method ' + name + ' not found in the reference source');
  let i = source.indexOf('{', source.indexOf(')', m.index));
  // the parameter list may itself contain braces only in destructuring, which these methods do not use
  let depth = 0, j = i;
  for (; j < source.length; j++) {
    const c = source[j];
    if (c === '{') depth++;
    else if (c === '}') { depth--; if (depth === 0) break; }
  }
  return source.slice(m.index, j + 1);
}

function stripTypes(ts) {
  let js = ts;
  js = js.replace(/\b(private|public|protected)\s+/g, '');
  // const x: T = ...
  js = js.replace(/\b(const|let|var)\s+([A-Za-z_$][\w$]*)\s*:\s*[^=;\n]+?\s*=/g, '$1 $2 =');
  // signatures: name(p: T, q: U): R {
  js = js.replace(/([A-Za-z_$][\w$]*)\s*\(([^()]*)\)\s*(:\s*[A-Za-z_$][\w$.\[\]<>]*)?\s*\{/g, (all, fname, params, ret) => {
    if (['if', 'for', 'while', 'switch', 'catch'].includes(fname)) return all;
    if (!ret && !/:/.test(params)) return all;
    const stripped = params.split(',').map((p) => p.replace(/\s*:\s*[\s\S]*$/, '').trim()).filter((p) => p.length).join(', ');
    return fname + '(' + stripped + ') {';
  });
  js = js.replace(/null!/g, 'null').replace(/\)!/g, ')');
  return js;
}

// ------------------------------------------------------------------ THREE shim (three@0.171.0)

class Vector3 {
  constructor(x, y, z) { this.x = x || 0; this.y = y || 0; this.z = z || 0; }
  set(x, y, z) { this.x = x; this.y = y; this.z = z; return this; }
  min(v) { this.x = Math.min(this.x, v.x); this.y = Math.min(this.y, v.y); this.z = Math.min(this.z, v.z); return this; }
  max(v) { this.x = Math.max(this.x, v.x); this.y = Math.max(this.y, v.y); this.z = Math.max(this.z, v.z); return this; }
  addVectors(a, b) { this.x = a.x + b.x; this.y = a.y + b.y; this.z = a.z + b.z; return this; }
  subVectors(a, b) { this.x = a.x - b.x; this.y = a.y - b.y; this.z = a.z - b.z; return this; }
  multiplyScalar(s) { this.x *= s; this.y *= s; this.z *= s; return this; }
  toArray() { return [this.x, this.y, this.z]; }
}

class Box3 {
  constructor() {
    this.min = new Vector3(+Infinity, +Infinity, +Infinity);
    this.max = new Vector3(-Infinity, -Infinity, -Infinity);
  }
  makeEmpty() { this.min.set(+Infinity, +Infinity, +Infinity); this.max.set(-Infinity, -Infinity, -Infinity); return this; }
  isEmpty() { return (this.max.x < this.min.x) || (this.max.y < this.min.y) || (this.max.z < this.min.z); }
  setFromPoints(points) { this.makeEmpty(); for (let i = 0, il = points.length; i < il; i++) this.expandByPoint(points[i]); return this; }
  expandByPoint(point) { this.min.min(point); this.max.max(point); return this; }
  getCenter(target) { return this.isEmpty() ? target.set(0, 0, 0) : target.addVectors(this.min, this.max).multiplyScalar(0.5); }
  getSize(target) { return this.isEmpty() ? target.set(0, 0, 0) : target.subVectors(this.max, this.min); }
}

const THREE = { Vector3, Box3, FloatType: 1015 };

function compile(methods) {
  const body = 'return class Extracted {\n' + methods.join('\n\n') + '\n};';
  return new Function('THREE', body)(THREE);      // eslint-disable-line no-new-func
}

// ------------------------------------------------------------------ drivers

function runBvh(root, positionsFile, outFile) {
  const src = fs.readFileSync(path.join(root, 'src', 'passes', 'raytrace.ts'), 'utf8');
  const Extracted = compile(['buildBVH', 'buildBVHRecursive', 'flattenBVH'].map((n) => stripTypes(extractMethod(src, n))));
  const raw = fs.readFileSync(positionsFile);
  const pos = new Float64Array(raw.buffer, raw.byteOffset, raw.length / 8);
  const triangles = [];
  for (let i = 0; i < pos.length; i += 9) {
    triangles.push({
      aPosition: new Vector3(pos[i], pos[i + 1], pos[i + 2]),
      bPosition: new Vector3(pos[i + 3], pos[i + 4], pos[i + 5]),
      cPosition: new Vector3(pos[i + 6], pos[i + 7], pos[i + 8]),
    });
  }
  const pass = new Extracted();
  const flat = pass.flattenBVH(pass.buildBVH(triangles));
  // updateBVHBuffer (raytrace.ts:177-193): webgpu-utils stores the fields into typed-array views
  const out = Buffer.alloc(flat.length * 48);
  const f32 = new Float32Array(out.buffer, out.byteOffset, out.length / 4);
  const i32 = new Int32Array(out.buffer, out.byteOffset, out.length / 4);
  flat.forEach((n, k) => {
    f32.set(n.min, 12 * k);
    f32.set(n.max, 12 * k + 4);
    i32[12 * k + 7] = n.isLeaf;
    i32[12 * k + 8] = n.left;
    i32[12 * k + 9] = n.right;
    i32[12 * k + 10] = n.triangleIndex;
  });
  fs.writeFileSync(outFile, out);
  console.log(JSON.stringify({ triangles: triangles.length, nodes: flat.length }));
}

function runCdf(root, envFile, outFile) {
  const src = fs.readFileSync(path.join(root, 'src', 'renderer.ts'), 'utf8');
  const Extracted = compile([stripTypes(extractMethod(src, 'updateEnvironmentTexture'))]);
  const raw = fs.readFileSync(envFile);
  const data = new Float32Array(raw.buffer, raw.byteOffset, raw.length / 4);
  const writes = [];
  const renderer = new Extracted();
  renderer.environmentTexture = 'env';
  renderer.environmentCDFTexture = 'cdf';
  renderer.device = { queue: { writeTexture: (dst, bytes, layout, size) => writes.push({ dst: dst.texture, bytes, layout, size }) } };
  renderer.updateEnvironmentTexture({ image: { width: 1024, height: 512, data }, type: THREE.FloatType });
  const cdf = writes.find((w) => w.dst === 'cdf');
  const env = writes.find((w) => w.dst === 'env');
  if (!cdf || !env || env.bytes !== data) throw new Error('unexpected writeTexture calls');
  fs.writeFileSync(outFile, Buffer.from(cdf.bytes.buffer, cdf.bytes.byteOffset, cdf.bytes.byteLength));
  console.log(JSON.stringify({ writes: writes.length, bytesPerRow: cdf.layout.bytesPerRow, width: cdf.size.width, height: cdf.size.height }));
}

const [mode, root, input, output] = process.argv.slice(2);
if (mode === 'bvh') runBvh(root, input, output);
else if (mode === 'cdf') runCdf(root, input, output);
else { console.error('usage: node run_reference_host.js bvh|cdf <reference-root> <input> <output>'); process.exit(2); }
