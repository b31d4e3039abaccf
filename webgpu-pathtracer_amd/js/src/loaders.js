'use strict';
// Asset ingestion for the headless host -- what src/main.ts does through three.js add-ons:
//   * GLTFLoader  (main.ts:251-266): .glb / .gltf -> a node hierarchy of indexed meshes
//   * the model placement rule of main.ts:268-279 (position (0, 0.5, 0), uniform scale
//     1 / max(bounds.max), every mesh gets the white material)
//   * RGBELoader.setDataType(FloatType) (main.ts:41-46): Radiance .hdr -> RGBA float texels
//   * OBJ (not in the reference; a convenience for meshes that are not glTF)
// three@0.171.0 (GLTFLoader, RGBELoader, Box3, Matrix4.decompose) is outside the reference tree
// (yarn.lock:1380); the published algorithms are restated here in the same operation order, in
// JS doubles.  Draco-compressed primitives (the reference ships Google's wasm decoder under
// public/static/draco) are rejected with an error: out of scope for the hot path.
const fs = require('fs');
const path = require('path');
const { Vector3, Quaternion, Matrix4 } = require('./math3');
const { BufferAttribute, BufferGeometry, Object3D, Mesh, RaytracingMaterial, DataTexture, FloatType } = require('./scene');

// ------------------------------------------------------------------------------ glTF 2.0

const COMPONENT = {
  5120: [Int8Array, 1], 5121: [Uint8Array, 1], 5122: [Int16Array, 2], 5123: [Uint16Array, 2],
  5125: [Uint32Array, 4], 5126: [Float32Array, 4],
};
const TYPE_SIZE = { SCALAR: 1, VEC2: 2, VEC3: 3, VEC4: 4, MAT2: 4, MAT3: 9, MAT4: 16 };

function parseGLBContainer(buf) {
  const u8 = buf instanceof Uint8Array ? buf : new Uint8Array(buf);
  const dv = new DataView(u8.buffer, u8.byteOffset, u8.byteLength);
  if (dv.getUint32(0, true) !== 0x46546c67) throw new Error('not a binary glTF file (bad magic)');
  if (dv.getUint32(4, true) !== 2) throw new Error('unsupported glTF container version ' + dv.getUint32(4, true));
  const total = dv.getUint32(8, true);
  let off = 12, json = null, bin = null;
  while (off + 8 <= total) {
    const len = dv.getUint32(off, true), type = dv.getUint32(off + 4, true);
    const chunk = u8.subarray(off + 8, off + 8 + len);
    if (type === 0x4e4f534a) json = JSON.parse(Buffer.from(chunk).toString('utf8'));
    else if (type === 0x004e4942 && bin === null) bin = chunk;
    off += 8 + len + ((4 - (len & 3)) & 3);
  }
  if (!json) throw new Error('binary glTF file without a JSON chunk');
  return { json, bin };
}

class GLTFLoader {
  // data: Buffer/Uint8Array/ArrayBuffer of a .glb, or a .gltf JSON string/object; baseDir resolves
  // external buffer URIs.  Returns { scene, scenes } like GLTFLoader.parse's result.
  parse(data, baseDir) {
    let json, bin = null;
    if (typeof data === 'string') json = JSON.parse(data);
    else if (data instanceof ArrayBuffer || ArrayBuffer.isView(data)) {
      const u8 = data instanceof ArrayBuffer ? new Uint8Array(data) : new Uint8Array(data.buffer, data.byteOffset, data.byteLength);
      if (u8.length >= 4 && u8[0] === 0x67 && u8[1] === 0x6c && u8[2] === 0x54 && u8[3] === 0x46) ({ json, bin } = parseGLBContainer(u8));
      else json = JSON.parse(Buffer.from(u8).toString('utf8'));
    } else json = data;
    if (!json.asset || parseFloat(json.asset.version) < 2) throw new Error('Unsupported asset. glTF versions >=2.0 are supported.');
    const required = json.extensionsRequired || [];
    if (required.includes('KHR_draco_mesh_compression')) throw new Error('Draco-compressed glTF is not supported by this host');
    for (const ext of required) if (ext !== 'KHR_materials_emissive_strength') throw new Error('unsupported required glTF extension ' + ext);

    const buffers = (json.buffers || []).map((b, i) => {
      if (b.uri === undefined) {
        if (i !== 0 || !bin) throw new Error('glTF buffer ' + i + ' has no uri and there is no BIN chunk');
        return bin;
      }
      const m = /^data:[^;,]*(;base64)?,(.*)$/.exec(b.uri);
      if (m) return new Uint8Array(Buffer.from(m[1] ? m[2] : decodeURIComponent(m[2]), m[1] ? 'base64' : 'utf8'));
      if (baseDir === undefined) throw new Error('glTF buffer ' + i + ' refers to an external file but no base directory was given');
      return new Uint8Array(fs.readFileSync(path.resolve(baseDir, decodeURIComponent(b.uri))));
    });

    const accessor = (index) => {
      const a = json.accessors[index];
      if (a.sparse) throw new Error('sparse accessors are not supported');
      const [Ctor, csize] = COMPONENT[a.componentType] || [];
      if (!Ctor) throw new Error('bad accessor componentType ' + a.componentType);
      const n = TYPE_SIZE[a.type];
      const out = new Ctor(a.count * n);
      if (a.bufferView === undefined) return { array: out, itemSize: n, normalized: !!a.normalized };
      const view = json.bufferViews[a.bufferView];
      const src = buffers[view.buffer];
      const base = (view.byteOffset || 0) + (a.byteOffset || 0);
      const stride = view.byteStride || n * csize;
      const dv = new DataView(src.buffer, src.byteOffset, src.byteLength);
      const get = { 5120: 'getInt8', 5121: 'getUint8', 5122: 'getInt16', 5123: 'getUint16', 5125: 'getUint32', 5126: 'getFloat32' }[a.componentType];
      for (let i = 0; i < a.count; i++) for (let k = 0; k < n; k++) out[i * n + k] = dv[get](base + i * stride + k * csize, true);
      return { array: out, itemSize: n, normalized: !!a.normalized };
    };

    const white = new RaytracingMaterial();
    const buildPrimitive = (prim) => {
      if (prim.extensions && prim.extensions.KHR_draco_mesh_compression) throw new Error('Draco-compressed primitives are not supported by this host');
      if (prim.mode !== undefined && prim.mode !== 4) return null;           // TRIANGLES only
      if (prim.attributes.POSITION === undefined) return null;
      const pos = accessor(prim.attributes.POSITION);
      if (!(pos.array instanceof Float32Array) || pos.itemSize !== 3) throw new Error('POSITION must be float VEC3 (KHR_mesh_quantization is not supported)');
      const geometry = new BufferGeometry();
      geometry.setAttribute('position', new BufferAttribute(pos.array, 3));
      if (prim.indices !== undefined) geometry.setIndex(Uint32Array.from(accessor(prim.indices).array));
      if (prim.attributes.NORMAL !== undefined) {
        const nrm = accessor(prim.attributes.NORMAL);
        if (!(nrm.array instanceof Float32Array)) throw new Error('NORMAL must be float VEC3');
        geometry.setAttribute('normal', new BufferAttribute(nrm.array, 3));
      } else {
        // GLTFLoader leaves such geometry without normals (flat shading); the reference's flatten
        // would then fail on the missing attribute, so area-weighted vertex normals
        // (BufferGeometry.computeVertexNormals) are provided instead.
        computeVertexNormals(geometry);
      }
      return new Mesh(geometry, white);
    };

    const buildNode = (index) => {
      const def = json.nodes[index];
      let node;
      if (def.mesh !== undefined) {
        const prims = json.meshes[def.mesh].primitives.map(buildPrimitive).filter((m) => m !== null);
        if (prims.length === 1) node = prims[0];
        else { node = new Object3D(); prims.forEach((m) => node.add(m)); }      // GLTFLoader: a Group of primitives
      } else node = new Object3D();
      node.name = def.name || '';
      if (def.matrix !== undefined) {
        const m = new Matrix4();
        m.elements = Array.from(def.matrix);
        decompose(m, node.position, node.quaternion, node.scale);                // Object3D.applyMatrix4 on an identity node
      } else {
        if (def.translation) node.position.set(def.translation[0], def.translation[1], def.translation[2]);
        if (def.rotation) { node.quaternion.x = def.rotation[0]; node.quaternion.y = def.rotation[1]; node.quaternion.z = def.rotation[2]; node.quaternion.w = def.rotation[3]; }
        if (def.scale) node.scale.set(def.scale[0], def.scale[1], def.scale[2]);
      }
      for (const c of def.children || []) node.add(buildNode(c));
      return node;
    };

    const scenes = (json.scenes || [{ nodes: (json.nodes || []).map((_, i) => i) }]).map((s) => {
      const group = new Object3D();
      for (const n of s.nodes || []) group.add(buildNode(n));
      return group;
    });
    return { scene: scenes[json.scene || 0], scenes, asset: json.asset };
  }

  load(file) { return this.parse(fs.readFileSync(file), path.dirname(file)); }
}

// three's Matrix4.decompose
function decompose(m, position, quaternion, scale) {
  const te = m.elements;
  let sx = new Vector3(te[0], te[1], te[2]).length();
  const sy = new Vector3(te[4], te[5], te[6]).length();
  const sz = new Vector3(te[8], te[9], te[10]).length();
  const det = determinant(te);
  if (det < 0) sx = -sx;
  position.set(te[12], te[13], te[14]);
  const r = new Matrix4();
  r.elements = te.slice();
  const isx = 1 / sx, isy = 1 / sy, isz = 1 / sz;
  r.elements[0] *= isx; r.elements[1] *= isx; r.elements[2] *= isx;
  r.elements[4] *= isy; r.elements[5] *= isy; r.elements[6] *= isy;
  r.elements[8] *= isz; r.elements[9] *= isz; r.elements[10] *= isz;
  quaternion.setFromRotationMatrix(r);
  scale.set(sx, sy, sz);
}

function determinant(te) {
  const n11 = te[0], n12 = te[4], n13 = te[8], n14 = te[12];
  const n21 = te[1], n22 = te[5], n23 = te[9], n24 = te[13];
  const n31 = te[2], n32 = te[6], n33 = te[10], n34 = te[14];
  const n41 = te[3], n42 = te[7], n43 = te[11], n44 = te[15];
  return (
    n41 * (+n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) +
    n42 * (+n11 * n23 * n34 - n11 * n24 * n33 + n14 * n21 * n33 - n13 * n21 * n34 + n13 * n24 * n31 - n14 * n23 * n31) +
    n43 * (+n11 * n24 * n32 - n11 * n22 * n34 - n14 * n21 * n32 + n12 * n21 * n34 + n14 * n22 * n31 - n12 * n24 * n31) +
    n44 * (-n13 * n22 * n31 - n11 * n23 * n32 + n11 * n22 * n33 + n13 * n21 * n32 - n12 * n21 * n33 + n12 * n23 * n31));
}

// BufferGeometry.computeVertexNormals for indexed (or not) float positions
function computeVertexNormals(geometry) {
  const pos = geometry.getAttribute('position').array;
  const nrm = new Float32Array(pos.length);
  const index = geometry.getIndex();
  const count = index ? index.array.length : pos.length / 3;
  const pA = new Vector3(), pB = new Vector3(), pC = new Vector3(), cb = new Vector3(), ab = new Vector3();
  for (let i = 0; i < count; i += 3) {
    const a = index ? index.array[i] : i, b = index ? index.array[i + 1] : i + 1, c = index ? index.array[i + 2] : i + 2;
    pA.set(pos[3 * a], pos[3 * a + 1], pos[3 * a + 2]);
    pB.set(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
    pC.set(pos[3 * c], pos[3 * c + 1], pos[3 * c + 2]);
    cb.subVectors(pC, pB);
    ab.subVectors(pA, pB);
    cb.crossVectors(cb, ab);
    for (const v of [a, b, c]) {               // Float32Array accumulation, like the attribute in three
      nrm[3 * v] += cb.x; nrm[3 * v + 1] += cb.y; nrm[3 * v + 2] += cb.z;
    }
  }
  const n = new Vector3();
  for (let v = 0; v < nrm.length / 3; v++) {
    n.set(nrm[3 * v], nrm[3 * v + 1], nrm[3 * v + 2]).normalize();
    nrm[3 * v] = n.x; nrm[3 * v + 1] = n.y; nrm[3 * v + 2] = n.z;
  }
  geometry.setAttribute('normal', new BufferAttribute(nrm, 3));
  return geometry;
}

// Box3.setFromObject(object) (precise = false): union over meshes of the geometry's local bounding
// box with its 8 corners transformed by matrixWorld.
function boundsOfObject(object) {
  object.updateMatrixWorld(true);
  const min = new Vector3(Infinity, Infinity, Infinity), max = new Vector3(-Infinity, -Infinity, -Infinity);
  object.traverse((o) => {
    if (!(o instanceof Mesh) || !o.geometry) return;
    const p = o.geometry.getAttribute('position').array;
    let lx = Infinity, ly = Infinity, lz = Infinity, hx = -Infinity, hy = -Infinity, hz = -Infinity;
    for (let i = 0; i < p.length; i += 3) {
      if (p[i] < lx) lx = p[i]; if (p[i] > hx) hx = p[i];
      if (p[i + 1] < ly) ly = p[i + 1]; if (p[i + 1] > hy) hy = p[i + 1];
      if (p[i + 2] < lz) lz = p[i + 2]; if (p[i + 2] > hz) hz = p[i + 2];
    }
    if (lx > hx) return;
    for (const x of [lx, hx]) for (const y of [ly, hy]) for (const z of [lz, hz]) {     // Box3.applyMatrix4's corner order
      const c = new Vector3(x, y, z).applyMatrix4(o.matrixWorld);
      if (c.x < min.x) min.x = c.x; if (c.y < min.y) min.y = c.y; if (c.z < min.z) min.z = c.z;
      if (c.x > max.x) max.x = c.x; if (c.y > max.y) max.y = c.y; if (c.z > max.z) max.z = c.z;
    }
  });
  return { min, max };
}

// main.ts:49-53: the material every loaded mesh gets
function whiteMaterial() {
  const white = new RaytracingMaterial();
  white.color.set(1.0, 1.0, 1.0);
  white.roughness = 1;
  white.metalness = 0.02;
  white.specularColor.set(1.0, 1.0, 1.0);
  return white;
}

// main.ts:268-279
function placeModel(model, material) {
  if (material === undefined) material = whiteMaterial();
  model.position.x = 0;
  model.position.y = 0.5;
  model.position.z = 0;
  const bounds = boundsOfObject(model);
  const scale = 1 / Math.max(bounds.max.x, bounds.max.y, bounds.max.z);
  model.scale.set(scale, scale, scale);
  model.traverse((child) => { if (child instanceof Mesh) child.material = material; });
  return model;
}

// ------------------------------------------------------------------------------ OBJ

function parseOBJ(text) {
  const vs = [], vns = [], keyToIndex = new Map(), pos = [], nrm = [], index = [];
  let haveNormals = true;
  const vertex = (token) => {
    const parts = token.split('/');
    let vi = parseInt(parts[0], 10), ni = parts.length > 2 && parts[2] !== '' ? parseInt(parts[2], 10) : 0;
    if (vi < 0) vi = vs.length / 3 + 1 + vi;
    if (ni < 0) ni = vns.length / 3 + 1 + ni;
    if (ni === 0) haveNormals = false;
    const key = vi + '/' + ni;
    let id = keyToIndex.get(key);
    if (id === undefined) {
      id = pos.length / 3;
      keyToIndex.set(key, id);
      pos.push(vs[3 * (vi - 1)], vs[3 * (vi - 1) + 1], vs[3 * (vi - 1) + 2]);
      if (ni > 0) nrm.push(vns[3 * (ni - 1)], vns[3 * (ni - 1) + 1], vns[3 * (ni - 1) + 2]); else nrm.push(0, 0, 0);
    }
    return id;
  };
  for (const raw of text.split('\n')) {
    const line = raw.trim();
    if (line.length === 0 || line[0] === '#') continue;
    const t = line.split(/\s+/);
    if (t[0] === 'v') vs.push(parseFloat(t[1]), parseFloat(t[2]), parseFloat(t[3]));
    else if (t[0] === 'vn') vns.push(parseFloat(t[1]), parseFloat(t[2]), parseFloat(t[3]));
    else if (t[0] === 'f') {
      const ids = t.slice(1).map(vertex);
      for (let i = 1; i + 1 < ids.length; i++) index.push(ids[0], ids[i], ids[i + 1]);       // fan
    }
  }
  const geometry = new BufferGeometry();
  geometry.setAttribute('position', new BufferAttribute(new Float32Array(pos), 3));
  geometry.setIndex(index);
  if (haveNormals && nrm.length) geometry.setAttribute('normal', new BufferAttribute(new Float32Array(nrm), 3));
  else computeVertexNormals(geometry);
  const group = new Object3D();
  group.add(new Mesh(geometry, new RaytracingMaterial()));
  return group;
}

// ------------------------------------------------------------------------------ Radiance .hdr (RGBE)

// RGBELoader.parse + the FloatType conversion (RGBEByteToRGBFloat: v * 2^(e-128) / 255, alpha 1).
// Returns a DataTexture { image: { data: Float32Array RGBA, width, height }, type: FloatType };
// rows are in file order (-Y: top row first), which is what the reference uploads (renderer.ts:145-157).
function parseRGBE(buf) {
  const u8 = buf instanceof Uint8Array ? buf : new Uint8Array(buf);
  let p = 0;
  const readLine = () => {
    let s = '';
    while (p < u8.length) { const c = u8[p++]; if (c === 0x0a) return s; s += String.fromCharCode(c); }
    return p <= u8.length && s.length ? s : null;
  };
  const first = readLine();
  if (first === null || !/^#\?(\S+)/.test(first)) throw new Error('THREE.RGBELoader: Bad File Format: bad initial token');
  let format = false, width = 0, height = 0;
  for (;;) {
    const line = readLine();
    if (line === null) throw new Error('THREE.RGBELoader: Bad File Format: no header found');
    if (/^\s*FORMAT=(\S+)\s*$/.test(line)) format = true;
    const m = /^\s*-Y\s+(\d+)\s+\+X\s+(\d+)\s*$/.exec(line);
    if (m) { height = parseInt(m[1], 10); width = parseInt(m[2], 10); break; }
  }
  if (!format) throw new Error('THREE.RGBELoader: Bad File Format: missing format specifier');
  const rgbe = new Uint8Array(width * height * 4);
  if (width < 8 || width > 0x7fff || u8[p] !== 2 || u8[p + 1] !== 2 || (u8[p + 2] & 0x80)) {
    rgbe.set(u8.subarray(p, p + rgbe.length));                 // flat (not run-length encoded)
    if (u8.length - p < rgbe.length) throw new Error('THREE.RGBELoader: Read Error: not enough pixel data');
  } else {
    const scan = new Uint8Array(4 * width);
    for (let y = 0; y < height; y++) {
      if (p + 4 > u8.length) throw new Error('THREE.RGBELoader: Read Error');
      if (u8[p] !== 2 || u8[p + 1] !== 2 || ((u8[p + 2] << 8) | u8[p + 3]) !== width) throw new Error('THREE.RGBELoader: Bad File Format: bad rgbe scanline format');
      p += 4;
      let ptr = 0;
      while (ptr < 4 * width && p < u8.length) {
        let count = u8[p++];
        const run = count > 128;
        if (run) count -= 128;
        if (count === 0 || ptr + count > 4 * width) throw new Error('THREE.RGBELoader: Bad File Format: bad scanline data');
        if (run) { const v = u8[p++]; scan.fill(v, ptr, ptr + count); ptr += count; }
        else { scan.set(u8.subarray(p, p + count), ptr); ptr += count; p += count; }
      }
      for (let x = 0; x < width; x++) {
        const o = 4 * (y * width + x);
        rgbe[o] = scan[x]; rgbe[o + 1] = scan[x + width]; rgbe[o + 2] = scan[x + 2 * width]; rgbe[o + 3] = scan[x + 3 * width];
      }
    }
  }
  const data = new Float32Array(width * height * 4);
  for (let i = 0; i < width * height; i++) {
    const e = rgbe[4 * i + 3];
    const scale = Math.pow(2.0, e - 128.0) / 255.0;
    data[4 * i] = rgbe[4 * i] * scale;
    data[4 * i + 1] = rgbe[4 * i + 1] * scale;
    data[4 * i + 2] = rgbe[4 * i + 2] * scale;
    data[4 * i + 3] = 1;
  }
  return new DataTexture(data, width, height, FloatType);
}

class RGBELoader {
  setDataType(type) { if (type !== FloatType) throw new Error('only FloatType is supported'); return this; }
  parse(buffer) { return parseRGBE(buffer); }
  load(file) { return parseRGBE(fs.readFileSync(file)); }
}

class OBJLoader {
  parse(text) { return parseOBJ(text); }
  load(file) { return parseOBJ(fs.readFileSync(file, 'utf8')); }
}

module.exports = { GLTFLoader, OBJLoader, RGBELoader, placeModel, whiteMaterial, boundsOfObject, computeVertexNormals, parseGLBContainer };
