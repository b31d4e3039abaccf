'use strict';
// src/renderer.ts without the DOM: same public API (create / diagnostic / render / update /
// setUniforms / reset / start / pause / resize / destroy / on / emit; frames,
// samplesPerFrame, scalingFactor, status, frame, progress, hasFramesToSample, timings,
// width / height / scaledWidth / scaledHeight / aspect), the GPUDevice replaced by a
// libmi3pt.so context reached through the N-API addon.  canvas / context / format have no
// headless meaning; readOutput / readAccumulation / readCanvas are the read-back a headless
// drop-in needs instead of canvas.toDataURL (src/main.ts:351-356).
const path = require('path');
const { RaytracePass } = require('./passes/raytrace');
const { AccumulatePass } = require('./passes/accumulate');
const { FullscreenPass } = require('./passes/fullscreen');
const { FloatType } = require('./scene');

const TEX_OUTPUT = 0, TEX_ACCUMULATION = 1, TEX_CANVAS = 2;

let nativeModule = null;
function loadNative() {
  if (nativeModule === null) {
    try {
      nativeModule = require(path.join(__dirname, '..', 'mi3pt.node'));
    } catch (err) {
      throw new Error('mi3pt.node / libmi3pt.so not loadable (build with __graft_entry__.build()): ' + err.message);
    }
  }
  return nativeModule;
}

class Renderer {
  constructor(args) {                                   // renderer.ts:47-92
    this.native = args.native;
    this.handle = args.handle;
    // presentEveryFrame: encode the fullscreen pass on every render() like renderer.ts:386 (false: never).
    // presentLatest: false (the default since round 4: the reference's canvas semantics, renderer.ts:379-390) -- every render()
    // gets its own accumulate and fullscreen pass and the canvas is drawn from the mean up to and including this very frame
    // (MI3PT_PRESENT_EXACT; up to 16 such frames share one raytrace launch, every canvas the reference would draw is drawn).
    // true (a headless program that only reads the final canvas): a sample frame that also presents is queued like any other
    // and the canvas is drawn once per launched batch (MI3PT_PRESENT_LATEST: 2.4x the frame rate); reading the canvas, or a
    // render() after sampling has stopped, shows every frame.
    this.options = Object.assign({ enableTimestampQuery: false, verbose: false, presentEveryFrame: true,
      presentLatest: false }, args.options || {});
    this.native.setPresentMode(this.handle, this.options.presentLatest ? 1 : 0);
    this._width = 0;
    this._height = 0;
    this._frame = 1;
    this._scalingFactor = 0.25;
    this.frames = 64;
    this.samplesPerFrame = 1;
    this.status = 'idle';
    this.listeners = new Map();
    this.tile = args.tile || { rank: 0, nranks: 1, blockRows: 8 };
    if (this.options.enableTimestampQuery) this.native.enableTiming(this.handle, 1);
    this.passes = {
      raytrace: new RaytracePass(this),
      accumulate: new AccumulatePass(this),
      fullscreen: new FullscreenPass(this),
    };
  }

  // ---- renderer.ts:470-533
  static async diagnostic() {
    let native;
    try { native = loadNative(); } catch (err) { return { supported: false }; }
    if (native.deviceCount() <= 0) return { supported: false };
    return { supported: true, info: { description: native.deviceName(0), vendor: 'amd', architecture: 'gfx950' } };
  }

  // options.device: one GPU (default 0).  options.devices = [0, 1, ...]: a device group (mi3pt_create_group) -- the
  // image's 8-row blocks are dealt to one member context per listed GPU, the scene is replicated, nothing is
  // exchanged per frame, and reading the accumulation image or the canvas gathers the members' rows with peer
  // copies; the loop below (renderer.ts:366-395) and every read-back stay the same and move whole images.
  static async create(options) {
    const native = loadNative();
    const opts = options || {};
    if (Array.isArray(opts.devices)) {
      if (opts.tile) throw new Error('Renderer.create: `devices` and `tile` exclude each other (a device group deals the tiles itself)');
      const handle = native.createGroup(opts.devices, opts.blockRows || 8);      // throws "HIP device not found."
      const r = new Renderer({ native, handle, options: opts, tile: { rank: 0, nranks: 1, blockRows: opts.blockRows || 8 } });
      r.devices = opts.devices.slice();
      return r;
    }
    const handle = native.create(opts.device || 0);      // throws "HIP device not found." (renderer.ts:514-516)
    const tile = opts.tile || { rank: 0, nranks: 1, blockRows: 8 };
    native.setTile(handle, tile.rank, tile.nranks, tile.blockRows);
    return new Renderer({ native, handle, options: opts, tile });
  }

  // ---- renderer.ts:132-281
  updateEnvironmentTexture(texture) {
    if (texture.image.width !== 1024 || texture.image.height !== 512) {
      throw new Error('Environment texture must be 1024x512 pixels. Please resize the texture and try again.');
    }
    if (texture.type !== FloatType) {
      throw new Error('Environment texture must be a floating point texture. Please convert the texture and try again.');
    }
    const data = texture.image.data;
    this.native.uploadEnvironment(this.handle, data, 1024, 512);
    this.native.uploadEnvironmentCdf(this.handle, this.native.hostEnvCdf(data, 1024, 512), 1024, 512);
  }

  // ---- renderer.ts:283-324
  resize(width, height) {
    if (this._width === width && this._height === height) return;
    this._width = width;
    this._height = height;
    this.native.resize(this.handle, width, height);
    this.reset();
    this.emit('resize');
  }
  get scalingFactor() { return this._scalingFactor; }
  set scalingFactor(value) {
    this._scalingFactor = value;
    this.setUniforms('fullscreen', { scalingFactor: value });
  }
  get width() { return this._width; }
  get scaledWidth() { return this._width * this._scalingFactor; }
  get height() { return this._height; }
  get scaledHeight() { return this._height * this._scalingFactor; }
  get aspect() { return this._width / this._height; }

  // ---- renderer.ts:330-356
  get hasFramesToSample() { return this._frame <= this.frames; }
  get progress() { return this._frame / (this.frames + 1); }
  get frame() { return this._frame; }
  set frame(value) {
    this._frame = value;
    if (this._frame > this.frames) {
      this.status = 'idle';
      this.emit('complete');
    }
  }
  get timings() {
    return {
      raytrace: this.passes.raytrace.timingAverage,
      accumulate: this.passes.accumulate.timingAverage,
      fullscreen: this.passes.fullscreen.timingAverage,
    };
  }

  setUniforms(pass, value) { this.passes[pass].setUniforms(value); }        // renderer.ts:358-360
  update(scene, camera) { this.passes.raytrace.updateScene(scene, camera); } // renderer.ts:362-364

  render(scene, camera) {                                                    // renderer.ts:366-395
    this.update(scene, camera);
    const shouldSample = this.status === 'sampling' && this.hasFramesToSample;
    if (shouldSample) this.frame++;
    this.passes.raytrace.update();
    this.passes.accumulate.update();
    this.passes.fullscreen.update();
    const commandEncoder = { passes: 0 };
    if (shouldSample) {
      this.passes.raytrace.render(commandEncoder);
      this.emit('progress', this.progress);
    }
    if (shouldSample) this.passes.accumulate.render(commandEncoder);
    if (this.options.presentEveryFrame) this.passes.fullscreen.render(commandEncoder);
    if (commandEncoder.passes) this.native.submit(this.handle, commandEncoder.passes);   // queue.submit
    if (shouldSample) this.passes.raytrace.updateTimings();
    if (shouldSample) this.passes.accumulate.updateTimings();
    if (this.options.presentEveryFrame) this.passes.fullscreen.updateTimings();
  }

  reset() {                                                                  // renderer.ts:397-416
    const prevStatus = this.status;
    this.status = 'paused';
    if (this._width > 0) this.native.reset(this.handle);
    this.emit('reset');
    this._frame = 1;
    this.status = prevStatus === 'idle' ? 'sampling' : prevStatus;
    if (this.status === 'sampling') this.emit('start');
  }

  async destroy() {                                                          // renderer.ts:418-429
    this.native.sync(this.handle);
    this.native.destroy(this.handle);
  }

  start() { this.status = this._frame > this.frames ? 'idle' : 'sampling'; } // renderer.ts:431-437
  pause() {                                                                  // renderer.ts:439-444
    if (this.status !== 'paused') {
      this.status = 'paused';
      this.emit('pause');
    }
  }
  on(event, callback) {                                                      // renderer.ts:446-458
    if (!this.listeners.has(event)) this.listeners.set(event, []);
    this.listeners.get(event).push(callback);
  }
  emit(event) {                                                              // renderer.ts:460-468
    const args = Array.prototype.slice.call(arguments, 1);
    const list = this.listeners.get(event);
    if (list) list.forEach((callback) => callback.apply(null, args));
  }

  // ---- headless read-back
  get localRows() { return this.native.tileLocalRows(this._height, this.tile.rank, this.tile.nranks, this.tile.blockRows); }
  readOutput() { return this.native.readTexture(this.handle, TEX_OUTPUT, this.localRows * this._width * 4); }
  readAccumulation() { return this.native.readTexture(this.handle, TEX_ACCUMULATION, this.localRows * this._width * 4); }
  readCanvasFloat() { return this.native.readTexture(this.handle, TEX_CANVAS, this._height * this._width * 4); }
  readCanvas() { return this.native.readCanvasRgba8(this.handle, this._height * this._width * 4); }
  counters() { return this.native.getCounters(this.handle); }
  // main.ts:351-356 (canvas.toDataURL("image/png")): the presented canvas as a PNG file
  screenshot(file) {
    const png = encodePNG(this.readCanvas(), this._width, this._height);
    if (file) require('fs').writeFileSync(file, png);
    return png;
  }
  // write the HDR accumulation image back (checkpoint/resume; a gathered multi-GPU image)
  writeAccumulation(data) { this.native.writeTexture(this.handle, TEX_ACCUMULATION, data); }
  // launch queued sample frames now without waiting for them (render() only queues them)
  flush() { this.native.flush(this.handle); }
  // queue.onSubmittedWorkDone() (renderer.ts:420), blocking: everything rendered so far has finished
  sync() { this.native.sync(this.handle); }
  // run the reference's dormant environment importance sampling (raytrace.wgsl:398-404 un-commented)
  setEnvironmentSampling(enabled) { this.native.setEnvSampling(this.handle, enabled ? 1 : 0); this.reset(); }
  raytraceLaunchStats(reset) { return this.native.raytraceLaunchStats(this.handle, reset ? 1 : 0); }
}

// Minimal PNG writer (8-bit RGBA, one IDAT, zlib from Node): what canvas.toDataURL("image/png") holds
function encodePNG(rgba, width, height) {
  const zlib = require('zlib');
  const crcTable = [];
  for (let n = 0; n < 256; n++) {
    let c = n;
    for (let k = 0; k < 8; k++) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1;
    crcTable[n] = c >>> 0;
  }
  const crc32 = (buf) => {
    let c = 0xffffffff;
    for (let i = 0; i < buf.length; i++) c = crcTable[(c ^ buf[i]) & 0xff] ^ (c >>> 8);
    return (c ^ 0xffffffff) >>> 0;
  };
  const chunk = (type, data) => {
    const body = Buffer.concat([Buffer.from(type, 'ascii'), data]);
    const out = Buffer.alloc(8 + data.length + 4);
    out.writeUInt32BE(data.length, 0);
    body.copy(out, 4);
    out.writeUInt32BE(crc32(body), 8 + data.length);
    return out;
  };
  const ihdr = Buffer.alloc(13);
  ihdr.writeUInt32BE(width, 0);
  ihdr.writeUInt32BE(height, 4);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  const raw = Buffer.alloc((width * 4 + 1) * height);
  for (let y = 0; y < height; y++) {
    raw[y * (width * 4 + 1)] = 0;
    Buffer.from(rgba.buffer, rgba.byteOffset + y * width * 4, width * 4).copy(raw, y * (width * 4 + 1) + 1);
  }
  return Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), chunk('IHDR', ihdr),
    chunk('IDAT', zlib.deflateSync(raw)), chunk('IEND', Buffer.alloc(0))]);
}

module.exports = { Renderer, loadNative, encodePNG };
