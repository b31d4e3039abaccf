'use strict';
// src/passes/raytrace.ts -- uniforms, scene flattening and upload.  BVH construction
// (buildBVH / buildBVHRecursive / flattenBVH, :540-694) is delegated to the native builder
// (mi3pt_host_build_bvh_f64), which produces the identical tree from the same doubles.
const { Pass } = require('./pass');
const { StructuredView } = require('../layout');
const { Vector3, Matrix3 } = require('../math3');
const { Mesh, RaytracingMaterial } = require('../scene');

const PASS_RAYTRACE = 0, SUBMIT_RAYTRACE = 1;

class RaytracePass extends Pass {
  constructor(renderer) {
    super(renderer);
    this.passId = PASS_RAYTRACE;
    this.uniforms = new StructuredView('RaytraceUniforms');
    this.stats = { Triangles: 0, Materials: 0, 'BVH Nodes': 0 };
  }

  setUniforms(value) {                       // raytrace.ts:359-369
    this.uniforms.set(value);
    this.renderer.native.setUniforms(this.renderer.handle, this.passId, this.uniforms.bytes);
  }

  update() {                                 // raytrace.ts:371-378
    this.setUniforms({
      resolution: [this.renderer.scaledWidth, this.renderer.scaledHeight],
      aspect: this.renderer.aspect,
      frame: this.renderer.frame,
      samplesPerFrame: this.renderer.samplesPerFrame,
    });
  }

  // raytrace.ts:406-502: indexed meshes -> world-space triangles + de-duplicated materials
  static flattenScene(scene) {
    scene.updateMatrixWorld(true);
    const meshes = [];
    scene.traverse((object) => {
      if (object instanceof Mesh && object.visible && object.material instanceof RaytracingMaterial) {
        meshes.push(object);
      }
    });
    const triangles = [];
    const materials = [];
    for (const mesh of meshes) {
      const indices = mesh.geometry.getIndex();
      const positions = mesh.geometry.getAttribute('position');
      const normals = mesh.geometry.getAttribute('normal');
      if (!indices) {
        console.warn('Mesh does not have indices');      // raytrace.ts:499-501
        continue;
      }
      const normalMatrix = new Matrix3().getNormalMatrix(mesh.matrixWorld);
      for (let i = 0; i < indices.array.length; i += 3) {
        const tri = { p: [], n: [] };
        for (let k = 0; k < 3; k++) {
          const vi = indices.array[i + k];
          tri.p.push(new Vector3().fromBufferAttribute(positions, vi).applyMatrix4(mesh.matrixWorld));
          tri.n.push(new Vector3().fromBufferAttribute(normals, vi).applyMatrix3(normalMatrix).normalize());
        }
        let materialIndex = materials.indexOf(mesh.material);
        if (materialIndex === -1) {
          materialIndex = materials.length;
          materials.push(mesh.material);
        }
        tri.materialIndex = materialIndex;
        triangles.push(tri);
      }
    }
    return { triangles, materials };
  }

  // raytrace.ts:104-121 / :138-160: structured views -> bytes
  static packScene(flat) {
    const { triangles, materials } = flat;
    const triView = new StructuredView('Triangle', Math.max(triangles.length, 1));
    const positions = new Float64Array(triangles.length * 9);
    for (let i = 0; i < triangles.length; i++) {
      const t = triangles[i];
      triView.set(i, {
        aPosition: t.p[0].toArray(), bPosition: t.p[1].toArray(), cPosition: t.p[2].toArray(),
        aNormal: t.n[0].toArray(), bNormal: t.n[1].toArray(), cNormal: t.n[2].toArray(),
        materialIndex: t.materialIndex,
      });
      for (let k = 0; k < 3; k++) {
        positions[9 * i + 3 * k] = t.p[k].x;
        positions[9 * i + 3 * k + 1] = t.p[k].y;
        positions[9 * i + 3 * k + 2] = t.p[k].z;
      }
    }
    const matView = new StructuredView('Material', Math.max(materials.length, 1));
    for (let i = 0; i < materials.length; i++) {
      const m = materials[i];
      matView.set(i, {
        color: m.color.toArray(), specularColor: m.specularColor.toArray(), roughness: m.roughness,
        metalness: m.metalness, emissionColor: m.emissive.toArray(), emissionStrength: m.emissiveIntensity,
      });
    }
    return { triangleBytes: triView.bytes, materialBytes: matView.bytes, positions };
  }

  updateScene(scene, camera) {               // raytrace.ts:380-533
    this.setUniforms({
      camera: {
        position: camera.getWorldPosition(new Vector3()).toArray(),
        direction: camera.getWorldDirection(new Vector3()).toArray(),
        fov: camera.fov,
        focalDistance: camera.focalDistance,
        aperture: camera.aperture,
      },
    });
    if (!scene.needsUpdate) return;

    const environment = scene.environment;
    if (environment) this.renderer.updateEnvironmentTexture(environment);

    const flat = RaytracePass.flattenScene(scene);
    if (flat.triangles.length === 0) throw new Error('Input nodes array is empty');    // raytrace.ts:563-565
    const packed = RaytracePass.packScene(flat);
    const native = this.renderer.native, handle = this.renderer.handle;
    let nodeBytes;
    if (this.renderer.options.deviceBvh) {
      // a linear BVH built on the GPU from the uploaded triangles (milliseconds on millions of triangles)
      // instead of the reference's SAH tree -- same closest hits, more box tests per ray (csrc/pt_lbvh.hip)
      native.uploadTriangles(handle, packed.triangleBytes);
      nodeBytes = native.deviceBuildBvh(handle, flat.triangles.length);
      native.uploadBvh(handle, nodeBytes);
    } else {
      nodeBytes = native.hostBuildBvhF64(packed.positions, this.renderer.options.builderThreads || 0);
      native.uploadBvh(handle, nodeBytes);
      native.uploadTriangles(handle, packed.triangleBytes);
    }
    native.uploadMaterials(handle, packed.materialBytes);
    scene.needsUpdate = false;
    this.stats = { Triangles: flat.triangles.length, Materials: flat.materials.length, 'BVH Nodes': nodeBytes.length / 48 };
    if (this.renderer.options.verbose) console.table(this.stats);                     // raytrace.ts:528-532
  }

  render(commandEncoder) { commandEncoder.passes |= SUBMIT_RAYTRACE; }     // raytrace.ts:696-708
}
module.exports = { RaytracePass };
