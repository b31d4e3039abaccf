'use strict';
// The reference's default scene and camera, src/main.ts:36-75: a 5x5 plane rotated -90 deg
// about X, a red 0.8^3 box at (0, 0.4, 0.5), a white sphere (r 0.5, 32x32) at (0, 0.5, -0.5);
// camera fov 45 at (0, 1, 4) looking at the origin (the OrbitControls target).
//
// NOTE: the material and mesh set-up below (from `const white` to the sphere being added) is a
// near-verbatim TRANSCRIPTION of src/main.ts:49-75 with `THREE.` replaced by `pt.` -- it is the
// reference's default scene used as an INPUT (test / benchmark data, the way main.ts hands it to
// its own renderer), not an implementation of anything.
const pt = require('..');

function buildDefaultScene(envData) {
  const scene = new pt.RaytracingScene();
  const camera = new pt.RaytracingCamera(45);
  camera.position.copy(new pt.Vector3(0, 1, 4));
  camera.lookAt(0, 0, 0);

  if (envData) {
    const env = new pt.DataTexture(envData, 1024, 512, pt.FloatType);
    scene.background = env;
    scene.environment = env;
  }

  const white = new pt.RaytracingMaterial();
  white.color.set(1.0, 1.0, 1.0);
  white.roughness = 1;
  white.metalness = 0.02;
  white.specularColor.set(1.0, 1.0, 1.0);

  const red = new pt.RaytracingMaterial();
  red.color.set(1.0, 0.05, 0.05);
  red.roughness = 1.0;
  red.metalness = 0.0;
  red.specularColor.set(1.0, 1.0, 1.0);

  const plane = new pt.Mesh(new pt.PlaneGeometry(5, 5), white);
  plane.rotateX(-Math.PI / 2);
  scene.add(plane);

  const box = new pt.Mesh(new pt.BoxGeometry(0.8, 0.8, 0.8), red);
  box.position.y = 0.4;
  box.position.z = 0.5;
  scene.add(box);

  const sphere = new pt.Mesh(new pt.SphereGeometry(0.5, 32, 32), white);
  sphere.position.y = 0.5;
  sphere.position.z = -0.5;
  scene.add(sphere);

  scene.needsUpdate = true;
  return { scene, camera };
}

// src/main.ts:83-92
const PARAMS = { maxBounces: 4, denoise: true, accumulate: true, tonemapping: 1, envMapIntensity: 1.0, envMapRotation: 0.0 };

module.exports = { buildDefaultScene, PARAMS };
