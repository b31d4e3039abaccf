'use strict';
// Byte layouts of the reference's WGSL structs, as webgpu-utils@1.0.2 computes them from
// the shader source (src/passes/raytrace.ts:53,89-94,123-128,162-167,195-197), and a
// StructuredView with the same *partial* set() semantics the reference relies on
// (src/renderer.ts:358-360): fields that are not named keep their value.
//
//   Triangle 112 B  raytrace.wgsl:40-49     BVHNode 48 B  raytrace.wgsl:51-64
//   Material  64 B  raytrace.wgsl:31-38     Uniforms 96 B raytrace.wgsl:66-75
//   accumulate Uniforms 16 B accumulate.wgsl:1-5 ; fullscreen Uniforms 24 B fullscreen.wgsl:14-20

const STRUCTS = {
  Triangle: {
    size: 112,
    fields: {
      aPosition: ['f32', 0, 3], bPosition: ['f32', 16, 3], cPosition: ['f32', 32, 3],
      aNormal: ['f32', 48, 3], bNormal: ['f32', 64, 3], cNormal: ['f32', 80, 3],
      materialIndex: ['i32', 92, 1], aabbIndex: ['i32', 96, 1],
    },
  },
  BVHNode: {
    size: 48,
    fields: {
      min: ['f32', 0, 3], max: ['f32', 16, 3], isLeaf: ['i32', 28, 1], left: ['i32', 32, 1],
      right: ['i32', 36, 1], triangleIndex: ['i32', 40, 1],
    },
  },
  Material: {
    size: 64,
    fields: {
      color: ['f32', 0, 3], specularColor: ['f32', 16, 3], roughness: ['f32', 28, 1],
      metalness: ['f32', 32, 1], emissionColor: ['f32', 48, 3], emissionStrength: ['f32', 60, 1],
    },
  },
  RaytraceUniforms: {
    size: 96,
    fields: {
      resolution: ['f32', 0, 2], aspect: ['f32', 8, 1], frame: ['u32', 12, 1],
      maxBounces: ['i32', 16, 1], samplesPerFrame: ['i32', 20, 1],
      camera: {
        position: ['f32', 32, 3], direction: ['f32', 48, 3], fov: ['f32', 60, 1],
        focalDistance: ['f32', 64, 1], aperture: ['f32', 68, 1],
      },
      envMapIntensity: ['f32', 80, 1], envMapRotation: ['f32', 84, 1],
    },
  },
  AccumulateUniforms: {
    size: 16,
    fields: { resolution: ['u32', 0, 2], frame: ['u32', 8, 1], enabled: ['u32', 12, 1] },
  },
  FullscreenUniforms: {
    size: 24,
    fields: {
      resolution: ['f32', 0, 2], aspect: ['f32', 8, 1], scalingFactor: ['f32', 12, 1],
      denoise: ['u32', 16, 1], tonemapping: ['u32', 20, 1],
    },
  },
};

function writeField(dv, base, spec, value) {
  const type = spec[0], offset = base + spec[1], count = spec[2];
  const values = (typeof value === 'number' || typeof value === 'boolean') ? [Number(value)] : value;
  for (let i = 0; i < count && i < values.length; i++) {
    const v = values[i];
    if (type === 'f32') dv.setFloat32(offset + 4 * i, v, true);
    else if (type === 'i32') dv.setInt32(offset + 4 * i, v, true);          // typed-array store: ToInt32
    else dv.setUint32(offset + 4 * i, v, true);                              // ToUint32 (truncates 22.5 -> 22)
  }
}

function setFields(dv, base, fields, value) {
  for (const key of Object.keys(value)) {
    const spec = fields[key];
    if (spec === undefined) continue;                // webgpu-utils ignores unknown keys
    if (Array.isArray(spec)) writeField(dv, base, spec, value[key]);
    else setFields(dv, base, spec, value[key]);      // nested struct (camera)
  }
}

class StructuredView {
  constructor(structName, count) {
    this.struct = STRUCTS[structName];
    if (!this.struct) throw new Error('unknown struct ' + structName);
    this.count = count === undefined ? 1 : count;
    this.arrayBuffer = new ArrayBuffer(this.struct.size * this.count);
    this.dataView = new DataView(this.arrayBuffer);
  }
  // set(partial) for a single struct, set(index, partial) for an array element
  set(a, b) {
    if (b === undefined) setFields(this.dataView, 0, this.struct.fields, a);
    else setFields(this.dataView, a * this.struct.size, this.struct.fields, b);
    return this;
  }
  get bytes() { return new Uint8Array(this.arrayBuffer); }
}

module.exports = { STRUCTS, StructuredView };
