"""Python mirror of the reference's Renderer / Pass API over the C ABI.

Same names, argument meaning and state machine as src/renderer.ts and src/passes/*.ts
so that tests read like the reference's call sites (src/main.ts:374-400):

    renderer = Renderer.create()
    renderer.resize(1920, 1080); renderer.scalingFactor = 1
    renderer.setUniforms("raytrace", {"maxBounces": 8, "envMapIntensity": 1.0})
    scene.needsUpdate = True
    renderer.render(scene, camera)          # one frame: raytrace + accumulate + fullscreen

Everything that computes pixels happens in libmi3pt.so on the GPU; this file only
keeps the host-side state the reference keeps in TypeScript (frame counter, uniform
views, events).  The Node host in webgpu-pathtracer_amd/js mirrors the same API in
JavaScript for the N-API path.
"""
import math

import numpy as np

from . import capi, layout


class RollingAverage:
    """timing.ts:1-20"""

    def __init__(self, num_samples=30):
        self.num_samples = num_samples
        self.samples = []
        self.cursor = 0
        self.total = 0.0

    def addSample(self, v):
        old = self.samples[self.cursor] if self.cursor < len(self.samples) else 0.0
        self.total += v - old                      # one rounding, like `total += v - (samples[cursor] || 0)`
        if self.cursor < len(self.samples):
            self.samples[self.cursor] = v
        else:
            self.samples.append(v)
        self.cursor = (self.cursor + 1) % self.num_samples

    @property
    def value(self):
        return self.total / len(self.samples) if self.samples else float("nan")     # 0 / 0 before the first sample


class RaytracingScene:
    """scene.ts:3-5 plus what RaytracePass.updateScene needs: a flattened scene
    (mi3pt_host.scenes.Scene) and an optional 1024x512 float environment."""

    def __init__(self, content=None, environment=None):
        self.content = content
        self.environment = environment
        self.needsUpdate = False


class RaytracingCamera:
    """scene.ts:7-10 (PerspectiveCamera subset: fov in degrees, position, target)"""

    def __init__(self, fov=45.0, position=(0.0, 1.0, 4.0), target=(0.0, 0.0, 0.0)):
        self.fov = fov
        self.position = tuple(position)
        self.target = tuple(target)
        self.focalDistance = 1.0
        self.aperture = 0.0

    def getWorldPosition(self):
        return self.position

    def getWorldDirection(self):
        d = np.array(self.target, np.float64) - np.array(self.position, np.float64)
        return tuple(d / math.sqrt(float(d @ d)))


class Pass:
    """pass.ts:4-27"""

    def __init__(self, renderer):
        self.renderer = renderer
        self.timingAverage = RollingAverage()

    def updateTimings(self):
        if not self.renderer.options["enableTimestampQuery"]:
            return
        try:
            us = self.renderer.ctx.pass_time_us(self.PASS_ID)
        except capi.Mi3ptError as e:
            if e.code == 4:       # pass not timed in the last submit (fused into the raytrace kernel)
                return
            raise
        self.timingAverage.addSample(us)


class RaytracePass(Pass):
    PASS_ID = capi.PASS_RAYTRACE

    def __init__(self, renderer):
        super().__init__(renderer)
        self.uniforms = layout.UniformBlock(layout.RAYTRACE_UNIFORMS)

    def setUniforms(self, value):                 # raytrace.ts:359-369
        self.uniforms.set(value)
        self.renderer.ctx.set_uniforms(self.PASS_ID, self.uniforms.tobytes())

    def update(self):                             # raytrace.ts:371-378
        r = self.renderer
        self.setUniforms({"resolution": [r.scaledWidth, r.scaledHeight], "aspect": r.aspect,
                          "frame": r.frame, "samplesPerFrame": r.samplesPerFrame})

    def updateScene(self, scene, camera):         # raytrace.ts:380-533
        self.setUniforms({"camera": {"position": camera.getWorldPosition(),
                                     "direction": camera.getWorldDirection(),
                                     "fov": camera.fov, "focalDistance": camera.focalDistance,
                                     "aperture": camera.aperture}})
        if not scene.needsUpdate:
            return
        if scene.environment is not None:
            self.renderer.updateEnvironmentTexture(scene.environment)
        content = scene.content
        if content is None or len(content.triangles) == 0:
            raise capi.Mi3ptError(1, "Input nodes array is empty")      # raytrace.ts:563-565
        if content.nodes is None:
            content.build_bvh()
        ctx = self.renderer.ctx
        ctx.upload_bvh(content.nodes)
        ctx.upload_triangles(content.triangles)
        ctx.upload_materials(content.material_bytes)
        scene.needsUpdate = False

    def render(self, encoder):                    # raytrace.ts:696-708
        encoder.append(capi.SUBMIT_RAYTRACE)


class AccumulatePass(Pass):
    PASS_ID = capi.PASS_ACCUMULATE

    def __init__(self, renderer):
        super().__init__(renderer)
        self.uniforms = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS)

    def setUniforms(self, value):                 # accumulate.ts:178-188
        self.uniforms.set(value)
        self.renderer.ctx.set_uniforms(self.PASS_ID, self.uniforms.tobytes())

    def update(self):                             # accumulate.ts:190-195
        r = self.renderer
        self.setUniforms({"resolution": [r.scaledWidth, r.scaledHeight], "frame": r.frame})

    def render(self, encoder):                    # accumulate.ts:154-176
        encoder.append(capi.SUBMIT_ACCUMULATE)


class FullscreenPass(Pass):
    PASS_ID = capi.PASS_FULLSCREEN

    def __init__(self, renderer):
        super().__init__(renderer)
        self.uniforms = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS)

    def setUniforms(self, value):                 # fullscreen.ts:138-148
        self.uniforms.set(value)
        self.renderer.ctx.set_uniforms(self.PASS_ID, self.uniforms.tobytes())

    def update(self):                             # fullscreen.ts:150-156
        r = self.renderer
        self.setUniforms({"resolution": [r.width, r.height], "aspect": r.aspect,
                          "scalingFactor": r.scalingFactor})

    def render(self, encoder):                    # fullscreen.ts:158-177
        encoder.append(capi.SUBMIT_FULLSCREEN)


class Renderer:
    """renderer.ts:20-468 without the DOM parts (canvas, ResizeObserver)."""

    def __init__(self, ctx, options=None):
        self.ctx = ctx
        # presentLatest False (the default since round 4): the reference's canvas semantics -- every render() that presents gets
        # its own accumulate and fullscreen pass (renderer.ts:379-390).  True (a headless program that only reads the final
        # canvas): the canvas is drawn once per launched batch instead of once per render(); reading it, or a render() after
        # sampling stopped, shows every frame
        self.options = {"enableTimestampQuery": False, "presentLatest": False}
        self.options.update(options or {})
        ctx.set_present_mode(capi.PRESENT_LATEST if self.options["presentLatest"] else capi.PRESENT_EXACT)
        self._width = self._height = 0
        self._frame = 1
        self._scalingFactor = 0.25
        self.frames = 64
        self.samplesPerFrame = 1
        self.status = "idle"
        self.listeners = {}
        self.presentEveryFrame = True      # headless hosts may skip the fullscreen pass
        self.passes = {"raytrace": RaytracePass(self), "accumulate": AccumulatePass(self),
                       "fullscreen": FullscreenPass(self)}
        if self.options["enableTimestampQuery"]:
            ctx.enable_timing(True)

    # ---- static constructors: renderer.ts:470-533 ----
    @staticmethod
    def diagnostic():
        try:
            n = capi.device_count()
        except capi.Mi3ptError:
            return {"supported": False}
        if n <= 0:
            return {"supported": False}
        return {"supported": True, "info": capi.device_name(0)}

    @staticmethod
    def create(device=0, options=None, tile=None, devices=None):
        """devices = [0, 1, ...]: a device group (mi3pt_create_group) -- the same Renderer on several GPUs of one
        node: tiles dealt in 8-row blocks, the scene replicated, one gather when an image is read."""
        if devices is not None:
            if tile is not None:
                raise ValueError("`devices` and `tile` exclude each other (a device group deals the tiles itself)")
            ctx = capi.Context(devices=list(devices))
        else:
            ctx = capi.Context(device)          # raises "HIP device not found." without a GPU
        if tile is not None:
            ctx.set_tile(*tile)
        return Renderer(ctx, options)

    # ---- sizes: renderer.ts:283-324 ----
    def resize(self, width, height):
        if self._width == width and self._height == height:
            return
        self._width, self._height = int(width), int(height)
        self.ctx.resize(self._width, self._height)
        self.reset()
        self.emit("resize")

    @property
    def scalingFactor(self):
        return self._scalingFactor

    @scalingFactor.setter
    def scalingFactor(self, value):
        self._scalingFactor = value
        self.setUniforms("fullscreen", {"scalingFactor": value})

    width = property(lambda self: self._width)
    height = property(lambda self: self._height)
    scaledWidth = property(lambda self: self._width * self._scalingFactor)
    scaledHeight = property(lambda self: self._height * self._scalingFactor)
    aspect = property(lambda self: self._width / self._height)

    # ---- frame counter: renderer.ts:330-348 ----
    @property
    def hasFramesToSample(self):
        return self._frame <= self.frames

    @property
    def progress(self):
        return self._frame / (self.frames + 1)

    @property
    def frame(self):
        return self._frame

    @frame.setter
    def frame(self, value):
        self._frame = value
        if self._frame > self.frames:
            self.status = "idle"
            self.emit("complete")

    @property
    def timings(self):
        return {k: p.timingAverage for k, p in self.passes.items()}

    def setUniforms(self, which, value):           # renderer.ts:358-360
        self.passes[which].setUniforms(value)

    def update(self, scene, camera):               # renderer.ts:362-364
        self.passes["raytrace"].updateScene(scene, camera)

    def render(self, scene, camera):               # renderer.ts:366-395
        self.update(scene, camera)
        should_sample = self.status == "sampling" and self.hasFramesToSample
        if should_sample:
            self.frame = self._frame + 1
        self.passes["raytrace"].update()
        self.passes["accumulate"].update()
        self.passes["fullscreen"].update()
        encoder = []
        if should_sample:
            self.passes["raytrace"].render(encoder)
            self.emit("progress", self.progress)
        if should_sample:
            self.passes["accumulate"].render(encoder)
        if self.presentEveryFrame:
            self.passes["fullscreen"].render(encoder)
        mask = 0
        for bit in encoder:
            mask |= bit
        if mask:
            self.ctx.submit(mask)                  # queue.submit, renderer.ts:389-390
        if should_sample:
            self.passes["raytrace"].updateTimings()
            self.passes["accumulate"].updateTimings()
        if self.presentEveryFrame:
            self.passes["fullscreen"].updateTimings()

    def reset(self):                               # renderer.ts:397-416
        prev = self.status
        self.status = "paused"
        if self._width:
            self.ctx.reset()
        self.emit("reset")
        self._frame = 1
        self.status = "sampling" if prev == "idle" else prev
        if self.status == "sampling":
            self.emit("start")

    def destroy(self):                             # renderer.ts:418-429
        self.ctx.sync()
        self.ctx.close()

    def start(self):                               # renderer.ts:431-437
        self.status = "idle" if self._frame > self.frames else "sampling"

    def pause(self):                               # renderer.ts:439-444
        if self.status != "paused":
            self.status = "paused"
            self.emit("pause")

    def on(self, event, callback):                 # renderer.ts:446-458
        self.listeners.setdefault(event, []).append(callback)

    def emit(self, event, *args):                  # renderer.ts:460-468
        for cb in self.listeners.get(event, []):
            cb(*args)

    def updateEnvironmentTexture(self, env):       # renderer.ts:132-281
        env = np.asarray(env)
        if env.shape[1] != 1024 or env.shape[0] != 512:
            raise capi.Mi3ptError(1, "Environment texture must be 1024x512 pixels. "
                                     "Please resize the texture and try again.")
        if env.dtype != np.float32:
            raise capi.Mi3ptError(1, "Environment texture must be a floating point texture. "
                                     "Please convert the texture and try again.")
        self.ctx.upload_environment(env)
        self.ctx.upload_environment_cdf(capi.host_env_cdf(env))

    # ---- headless read-back (no counterpart in the reference) ----
    def readOutput(self):
        return self.ctx.read_texture(capi.TEX_OUTPUT)

    def readAccumulation(self):
        return self.ctx.read_texture(capi.TEX_ACCUMULATION)

    def readCanvas(self):
        return self.ctx.read_canvas_rgba8()

    def screenshot(self, path=None):
        """main.ts:351-356 (canvas.toDataURL("image/png")): the presented canvas as PNG bytes."""
        png = encode_png(self.readCanvas())
        if path:
            with open(path, "wb") as f:
                f.write(png)
        return png


def encode_png(rgba8):
    """8-bit RGBA, one IDAT chunk."""
    import struct
    import zlib
    import numpy as np
    img = np.ascontiguousarray(rgba8, np.uint8)
    h, w = img.shape[:2]
    raw = np.zeros((h, w * 4 + 1), np.uint8)
    raw[:, 1:] = img.reshape(h, w * 4)

    def chunk(kind, data):
        body = kind + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes())) + chunk(b"IEND", b""))

