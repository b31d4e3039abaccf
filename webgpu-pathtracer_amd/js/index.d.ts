// Typed surface of the Node host (hand-written: there is no tsc in the build image).
// Names and semantics follow umar-ahmed/webgpu-pathtracer src/renderer.ts, src/passes/*.ts
// and src/scene.ts; browser-only members (canvas, context, format, device) are absent and
// read-back methods are added for headless use.

export class Vector3 {
  x: number; y: number; z: number;
  constructor(x?: number, y?: number, z?: number);
  set(x: number, y: number, z: number): this;
  copy(v: Vector3): this;
  toArray(): [number, number, number];
  normalize(): this;
}
export class Quaternion { x: number; y: number; z: number; w: number; }
export class Matrix4 { elements: number[]; }
export class Matrix3 { elements: number[]; getNormalMatrix(m: Matrix4): this; }
export class Color { r: number; g: number; b: number; set(r: number, g: number, b: number): this; toArray(): number[]; }

export const FloatType: number;
export class BufferAttribute { array: Float32Array | Uint32Array; itemSize: number; count: number; }
export class BufferGeometry {
  setIndex(indices: number[] | Uint32Array): this;
  getIndex(): BufferAttribute | null;
  setAttribute(name: string, attr: BufferAttribute): this;
  getAttribute(name: string): BufferAttribute;
}
export class PlaneGeometry extends BufferGeometry { constructor(width?: number, height?: number, widthSegments?: number, heightSegments?: number); }
export class BoxGeometry extends BufferGeometry { constructor(width?: number, height?: number, depth?: number, ws?: number, hs?: number, ds?: number); }
export class SphereGeometry extends BufferGeometry { constructor(radius?: number, widthSegments?: number, heightSegments?: number); }
export class Object3D {
  position: Vector3; quaternion: Quaternion; scale: Vector3; visible: boolean;
  matrix: Matrix4; matrixWorld: Matrix4; children: Object3D[];
  add(o: Object3D): this; remove(o: Object3D): this; clear(): this;
  rotateX(a: number): this; rotateY(a: number): this; rotateZ(a: number): this;
  updateMatrixWorld(force?: boolean): void;
  traverse(cb: (o: Object3D) => void): void;
}
export class Mesh extends Object3D { geometry: BufferGeometry; material: RaytracingMaterial; constructor(g: BufferGeometry, m: RaytracingMaterial); }
export class DataTexture { image: { data: Float32Array; width: number; height: number }; type: number; constructor(data: Float32Array, width: number, height: number, type?: number); }

/** src/scene.ts:3-5 */
export class RaytracingScene extends Object3D { needsUpdate: boolean; environment: DataTexture | null; background: DataTexture | null; }
/** src/scene.ts:7-10 */
export class RaytracingCamera extends Object3D {
  fov: number; focalDistance: number; aperture: number;
  constructor(fov?: number, aspect?: number, near?: number, far?: number);
  lookAt(x: number | Vector3, y?: number, z?: number): this;
  getWorldPosition(target: Vector3): Vector3;
  getWorldDirection(target: Vector3): Vector3;
}
/** src/scene.ts:12-14 */
export class RaytracingMaterial {
  color: Color; specularColor: Color; emissive: Color;
  roughness: number; metalness: number; emissiveIntensity: number;
}

export class RollingAverage { addSample(v: number): void; readonly value: number; }

/** src/passes/pass.ts:4-27 */
export abstract class Pass {
  timingAverage: RollingAverage;
  constructor(renderer: Renderer);
  abstract render(commandEncoder: { passes: number }): void;
  abstract update(): void;
  updateTimings(): void;
}
export class RaytracePass extends Pass {
  setUniforms(value: object): void; update(): void; render(e: { passes: number }): void;
  updateScene(scene: RaytracingScene, camera: RaytracingCamera): void;
  stats: { Triangles: number; Materials: number; 'BVH Nodes': number };
}
export class AccumulatePass extends Pass { setUniforms(value: object): void; update(): void; render(e: { passes: number }): void; }
export class FullscreenPass extends Pass { setUniforms(value: object): void; update(): void; render(e: { passes: number }): void; }

export type RendererEventType = 'start' | 'pause' | 'reset' | 'progress' | 'complete' | 'resize';
export interface RendererOptions {
  device?: number;
  enableTimestampQuery?: boolean;
  presentEveryFrame?: boolean;
  /** default false: the reference's canvas semantics (a fullscreen pass per render()); true: draw the canvas once per launched batch (headless programs that only read the final canvas) */
  presentLatest?: boolean;
  verbose?: boolean;
  builderThreads?: number;
  /** build a linear BVH on the GPU instead of the reference's SAH tree (fast on huge meshes, same closest hits) */
  deviceBvh?: boolean;
  /** multi-GPU tile split: this process renders the blockRows-row blocks dealt to `rank`, round after round, BACK AND FORTH
   *  (rank r owns block r of even rounds and block nranks - 1 - r of odd ones: include/mi3pt.h, mi3pt_set_tile).  Use
   *  native.tileGlobalRow(localRow, rank, nranks, blockRows) / native.tileOwner(y, nranks, blockRows) to de-interleave
   *  gathered rows -- never a copy of the formula (it changed with ABI 3) */
  tile?: { rank: number; nranks: number; blockRows: number };
  /** a device group inside this one process (mi3pt_create_group): the image's blockRows-row blocks are dealt to one
   *  member context per listed GPU; render(), the read-backs and the events are unchanged and move whole images */
  devices?: number[];
  blockRows?: number;
}
/** src/renderer.ts:20-468 */
export class Renderer {
  frames: number; samplesPerFrame: number; scalingFactor: number;
  status: 'idle' | 'sampling' | 'paused';
  readonly frame: number; readonly progress: number; readonly hasFramesToSample: boolean;
  readonly width: number; readonly height: number; readonly scaledWidth: number; readonly scaledHeight: number; readonly aspect: number;
  readonly timings: { raytrace: RollingAverage; accumulate: RollingAverage; fullscreen: RollingAverage };
  static diagnostic(): Promise<{ supported: false } | { supported: true; info: { description: string } }>;
  static create(options?: RendererOptions): Promise<Renderer>;
  render(scene: RaytracingScene, camera: RaytracingCamera): void;
  update(scene: RaytracingScene, camera: RaytracingCamera): void;
  setUniforms(pass: 'raytrace' | 'accumulate' | 'fullscreen', value: object): void;
  updateEnvironmentTexture(texture: DataTexture): void;
  resize(width: number, height: number): void;
  reset(): void; start(): void; pause(): void; destroy(): Promise<void>;
  on(event: RendererEventType, callback: (...args: any[]) => void): void;
  emit(event: RendererEventType, ...args: any[]): void;
  /** headless read-back: RGBA float, row 0 = bottom of the picture */
  readOutput(): Float32Array; readAccumulation(): Float32Array;
  /** the fullscreen pass's target, row 0 = top */
  readCanvasFloat(): Float32Array; readCanvas(): Uint8Array;
  counters(): { rays: number; boxTests: number; triTests: number; hits: number; misses: number; stackOverflows: number; pixels: number };
  /** src/main.ts:351-356: the presented canvas as PNG bytes (written to `file` when given) */
  screenshot(file?: string): Buffer;
  /** write the HDR accumulation image back (checkpoint / resume, gathered multi-GPU image) */
  writeAccumulation(data: Float32Array): void;
  /** launch the sample frames render() has queued, without waiting for them */
  flush(): void;
  raytraceLaunchStats(reset?: boolean): { totalMs: number; launches: number; frames: number };
  /** the reference's dormant environment importance sampling (raytrace.wgsl:398-404 un-commented); default off */
  setEnvironmentSampling(enabled: boolean): void;
}

/** src/main.ts:251-266 -- .glb / .gltf (no Draco) to a node hierarchy of indexed meshes */
export class GLTFLoader {
  parse(data: ArrayBuffer | Uint8Array | string | object, baseDir?: string): { scene: Object3D; scenes: Object3D[]; asset: object };
  load(file: string): { scene: Object3D; scenes: Object3D[]; asset: object };
}
export class OBJLoader { parse(text: string): Object3D; load(file: string): Object3D; }
/** src/main.ts:41-46 -- Radiance .hdr to RGBA float texels (FloatType) */
export class RGBELoader { setDataType(type: number): this; parse(buffer: ArrayBuffer | Uint8Array): DataTexture; load(file: string): DataTexture; }
/** the `white` material of src/main.ts:49-53 */
export function whiteMaterial(): RaytracingMaterial;
/** src/main.ts:268-279: position (0, 0.5, 0), uniform scale 1 / max(bounds.max), one material */
export function placeModel(model: Object3D, material?: RaytracingMaterial): Object3D;
export function boundsOfObject(object: Object3D): { min: Vector3; max: Vector3 };
export function encodePNG(rgba: Uint8Array, width: number, height: number): Buffer;
