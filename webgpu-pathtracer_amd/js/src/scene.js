'use strict';
// Scene types of the headless host.  RaytracingScene / RaytracingCamera /
// RaytracingMaterial keep the reference's names and extra fields (src/scene.ts:3-14); the
// three.js base classes they extend there (Scene, PerspectiveCamera, MeshStandardMaterial,
// Mesh, BufferGeometry and the Plane/Box/Sphere generators used by src/main.ts:61-73) are
// replaced by the minimal stand-ins below, because three@0.171.0 cannot be installed here
// (no network).  Vertex order, normals and index order follow three.js r171.
const { Vector3, Quaternion, Matrix4, Color } = require('./math3');

const FloatType = 1015;     // THREE.FloatType

class BufferAttribute {
  constructor(array, itemSize) {
    this.array = array;
    this.itemSize = itemSize;
    this.count = array.length / itemSize;
  }
}

class BufferGeometry {
  constructor() {
    this.index = null;
    this.attributes = {};
  }
  setIndex(indices) {
    this.index = new BufferAttribute(indices instanceof Uint32Array ? indices : new Uint32Array(indices), 1);
    return this;
  }
  getIndex() { return this.index; }
  setAttribute(name, attr) { this.attributes[name] = attr; return this; }
  getAttribute(name) { return this.attributes[name]; }
}

class PlaneGeometry extends BufferGeometry {
  constructor(width, height, widthSegments, heightSegments) {
    super();
    width = width === undefined ? 1 : width;
    height = height === undefined ? 1 : height;
    const gridX = Math.floor(widthSegments === undefined ? 1 : widthSegments);
    const gridY = Math.floor(heightSegments === undefined ? 1 : heightSegments);
    const widthHalf = width / 2, heightHalf = height / 2;
    const gridX1 = gridX + 1, gridY1 = gridY + 1;
    const segmentWidth = width / gridX, segmentHeight = height / gridY;
    const indices = [], vertices = [], normals = [];
    for (let iy = 0; iy < gridY1; iy++) {
      const y = iy * segmentHeight - heightHalf;
      for (let ix = 0; ix < gridX1; ix++) {
        const x = ix * segmentWidth - widthHalf;
        vertices.push(x, -y, 0);
        normals.push(0, 0, 1);
      }
    }
    for (let iy = 0; iy < gridY; iy++) {
      for (let ix = 0; ix < gridX; ix++) {
        const a = ix + gridX1 * iy;
        const b = ix + gridX1 * (iy + 1);
        const c = (ix + 1) + gridX1 * (iy + 1);
        const d = (ix + 1) + gridX1 * iy;
        indices.push(a, b, d);
        indices.push(b, c, d);
      }
    }
    this.setIndex(indices);
    this.setAttribute('position', new BufferAttribute(new Float32Array(vertices), 3));
    this.setAttribute('normal', new BufferAttribute(new Float32Array(normals), 3));
  }
}

class BoxGeometry extends BufferGeometry {
  constructor(width, height, depth, widthSegments, heightSegments, depthSegments) {
    super();
    width = width === undefined ? 1 : width;
    height = height === undefined ? 1 : height;
    depth = depth === undefined ? 1 : depth;
    const ws = Math.floor(widthSegments === undefined ? 1 : widthSegments);
    const hs = Math.floor(heightSegments === undefined ? 1 : heightSegments);
    const ds = Math.floor(depthSegments === undefined ? 1 : depthSegments);
    const indices = [], vertices = [], normals = [];
    let numberOfVertices = 0;
    const buildPlane = (u, v, w, udir, vdir, pw, ph, pd, gridX, gridY) => {
      const segmentWidth = pw / gridX, segmentHeight = ph / gridY;
      const widthHalf = pw / 2, heightHalf = ph / 2, depthHalf = pd / 2;
      const gridX1 = gridX + 1, gridY1 = gridY + 1;
      let vertexCounter = 0;
      const vector = { x: 0, y: 0, z: 0 };
      for (let iy = 0; iy < gridY1; iy++) {
        const y = iy * segmentHeight - heightHalf;
        for (let ix = 0; ix < gridX1; ix++) {
          const x = ix * segmentWidth - widthHalf;
          vector[u] = x * udir; vector[v] = y * vdir; vector[w] = depthHalf;
          vertices.push(vector.x, vector.y, vector.z);
          vector[u] = 0; vector[v] = 0; vector[w] = pd > 0 ? 1 : -1;
          normals.push(vector.x, vector.y, vector.z);
          vertexCounter += 1;
        }
      }
      for (let iy = 0; iy < gridY; iy++) {
        for (let ix = 0; ix < gridX; ix++) {
          const a = numberOfVertices + ix + gridX1 * iy;
          const b = numberOfVertices + ix + gridX1 * (iy + 1);
          const c = numberOfVertices + (ix + 1) + gridX1 * (iy + 1);
          const d = numberOfVertices + (ix + 1) + gridX1 * iy;
          indices.push(a, b, d);
          indices.push(b, c, d);
        }
      }
      numberOfVertices += vertexCounter;
    };
    buildPlane('z', 'y', 'x', -1, -1, depth, height, width, ds, hs);   // px
    buildPlane('z', 'y', 'x', 1, -1, depth, height, -width, ds, hs);   // nx
    buildPlane('x', 'z', 'y', 1, 1, width, depth, height, ws, ds);     // py
    buildPlane('x', 'z', 'y', 1, -1, width, depth, -height, ws, ds);   // ny
    buildPlane('x', 'y', 'z', 1, -1, width, height, depth, ws, hs);    // pz
    buildPlane('x', 'y', 'z', -1, -1, width, height, -depth, ws, hs);  // nz
    this.setIndex(indices);
    this.setAttribute('position', new BufferAttribute(new Float32Array(vertices), 3));
    this.setAttribute('normal', new BufferAttribute(new Float32Array(normals), 3));
  }
}

class SphereGeometry extends BufferGeometry {
  constructor(radius, widthSegments, heightSegments) {
    super();
    radius = radius === undefined ? 1 : radius;
    widthSegments = Math.max(3, Math.floor(widthSegments === undefined ? 32 : widthSegments));
    heightSegments = Math.max(2, Math.floor(heightSegments === undefined ? 16 : heightSegments));
    const phiStart = 0, phiLength = Math.PI * 2, thetaStart = 0, thetaLength = Math.PI;
    const thetaEnd = Math.min(thetaStart + thetaLength, Math.PI);
    let index = 0;
    const grid = [], indices = [], vertices = [], normals = [];
    const vertex = new Vector3(), normal = new Vector3();
    for (let iy = 0; iy <= heightSegments; iy++) {
      const verticesRow = [];
      const v = iy / heightSegments;
      for (let ix = 0; ix <= widthSegments; ix++) {
        const u = ix / widthSegments;
        vertex.x = -radius * Math.cos(phiStart + u * phiLength) * Math.sin(thetaStart + v * thetaLength);
        vertex.y = radius * Math.cos(thetaStart + v * thetaLength);
        vertex.z = radius * Math.sin(phiStart + u * phiLength) * Math.sin(thetaStart + v * thetaLength);
        vertices.push(vertex.x, vertex.y, vertex.z);
        normal.copy(vertex).normalize();
        normals.push(normal.x, normal.y, normal.z);
        verticesRow.push(index++);
      }
      grid.push(verticesRow);
    }
    for (let iy = 0; iy < heightSegments; iy++) {
      for (let ix = 0; ix < widthSegments; ix++) {
        const a = grid[iy][ix + 1], b = grid[iy][ix];
        const c = grid[iy + 1][ix], d = grid[iy + 1][ix + 1];
        if (iy !== 0 || thetaStart > 0) indices.push(a, b, d);
        if (iy !== heightSegments - 1 || thetaEnd < Math.PI) indices.push(b, c, d);
      }
    }
    this.setIndex(indices);
    this.setAttribute('position', new BufferAttribute(new Float32Array(vertices), 3));
    this.setAttribute('normal', new BufferAttribute(new Float32Array(normals), 3));
  }
}

const X_AXIS = new Vector3(1, 0, 0), Y_AXIS = new Vector3(0, 1, 0), Z_AXIS = new Vector3(0, 0, 1);

class Object3D {
  constructor() {
    this.parent = null;
    this.children = [];
    this.position = new Vector3();
    this.quaternion = new Quaternion();
    this.scale = new Vector3(1, 1, 1);
    this.up = new Vector3(0, 1, 0);
    this.matrix = new Matrix4();
    this.matrixWorld = new Matrix4();
    this.visible = true;
  }
  add(object) {
    if (object.parent) object.parent.remove(object);
    object.parent = this;
    this.children.push(object);
    return this;
  }
  remove(object) {
    const i = this.children.indexOf(object);
    if (i !== -1) { object.parent = null; this.children.splice(i, 1); }
    return this;
  }
  clear() {
    for (const c of this.children) c.parent = null;
    this.children.length = 0;
    return this;
  }
  rotateOnAxis(axis, angle) {
    this.quaternion.multiply(new Quaternion().setFromAxisAngle(axis, angle));
    return this;
  }
  rotateX(angle) { return this.rotateOnAxis(X_AXIS, angle); }
  rotateY(angle) { return this.rotateOnAxis(Y_AXIS, angle); }
  rotateZ(angle) { return this.rotateOnAxis(Z_AXIS, angle); }
  updateMatrix() { this.matrix.compose(this.position, this.quaternion, this.scale); }
  updateMatrixWorld() {
    this.updateMatrix();
    if (this.parent === null) this.matrixWorld.copy(this.matrix);
    else this.matrixWorld.multiplyMatrices(this.parent.matrixWorld, this.matrix);
    for (const c of this.children) c.updateMatrixWorld();
  }
  traverse(callback) {
    callback(this);
    for (const c of this.children) c.traverse(callback);
  }
  getWorldPosition(target) {
    this.updateMatrixWorld();
    const e = this.matrixWorld.elements;
    return target.set(e[12], e[13], e[14]);
  }
}

class Mesh extends Object3D {
  constructor(geometry, material) {
    super();
    this.geometry = geometry;
    this.material = material;
  }
}

// src/scene.ts:12-14 (+ the MeshStandardMaterial defaults the packer reads, raytrace.ts:138-153)
class RaytracingMaterial {
  constructor() {
    this.color = new Color(1, 1, 1);
    this.roughness = 1;
    this.metalness = 0;
    this.emissive = new Color(0, 0, 0);
    this.emissiveIntensity = 1;
    this.specularColor = new Color();
  }
}

// src/scene.ts:3-5
class RaytracingScene extends Object3D {
  constructor() {
    super();
    this.needsUpdate = false;
    this.environment = null;
    this.background = null;
  }
}

// src/scene.ts:7-10
class RaytracingCamera extends Object3D {
  constructor(fov, aspect, near, far) {
    super();
    this.fov = fov === undefined ? 50 : fov;
    this.aspect = aspect === undefined ? 1 : aspect;
    this.near = near === undefined ? 0.1 : near;
    this.far = far === undefined ? 2000 : far;
    this.focalDistance = 1;
    this.aperture = 0;
  }
  // Object3D.lookAt for cameras: the camera looks down its local -Z
  lookAt(x, y, z) {
    const target = x instanceof Vector3 ? x : new Vector3(x, y, z);
    this.updateMatrixWorld();
    const position = new Vector3().set(this.matrixWorld.elements[12], this.matrixWorld.elements[13],
      this.matrixWorld.elements[14]);
    const m = new Matrix4().lookAt(position, target, this.up);
    this.quaternion.setFromRotationMatrix(m);
    return this;
  }
  getWorldDirection(target) {
    this.updateMatrixWorld();
    const e = this.matrixWorld.elements;
    return target.set(-e[8], -e[9], -e[10]).normalize();
  }
}

// what RGBELoader.setDataType(FloatType) hands to scene.environment (src/main.ts:41-46)
class DataTexture {
  constructor(data, width, height, type) {
    this.image = { data, width, height };
    this.type = type === undefined ? FloatType : type;
  }
}

module.exports = {
  FloatType, BufferAttribute, BufferGeometry, PlaneGeometry, BoxGeometry, SphereGeometry,
  Object3D, Mesh, RaytracingMaterial, RaytracingScene, RaytracingCamera, DataTexture,
};
