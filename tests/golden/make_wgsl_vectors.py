#!/usr/bin/env python3
"""Generates tests/golden/wgsl_vectors.npz by EXECUTING the reference's shader text.

The three WGSL files of the reference (src/passes/shaders/{raytrace,accumulate,fullscreen}.wgsl)
are read in place from the reference checkout and run through oracle/wgsl_interp.py on seeded
inputs; inputs and outputs are stored as plain arrays.  Nothing of the reference's source ends up
in the repository -- only the vectors and this script.  tests/test_wgsl_vectors.py then holds the
C oracle (and, on a GPU, the HIP kernels) to these vectors bit for bit.

Implementation-defined pieces are supplied as DESIGN.md "Pinned arithmetic" fixes them: the
transcendental functions come from the pinned polynomial implementations (oracle/libptoracle.so,
orc_math), the samplers / texture formats from the host code that creates them
(renderer.ts:77-85 linear clamp-to-edge for the environment, fullscreen.ts:49-57 linear repeat),
the fragment shader's interpolated uv from the quad's corner values.

usage: python tests/golden/make_wgsl_vectors.py [/root/reference] [out.npz]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "webgpu-pathtracer_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pt_oracle as orc            # noqa: E402  (pinned math only)
from oracle import wgsl_interp as wi           # noqa: E402
from mi3pt_host import layout, scenes          # noqa: E402

F32 = np.float32


def pinned_math():
    one = lambda fn: (lambda x: orc.math_fn(fn, np.array([x], F32))[0])
    two = lambda fn: (lambda a, b: orc.math_fn(fn, np.array([a], F32), np.array([b], F32))[0])
    return {"sin": one(0), "cos": one(1), "tan": one(2), "log": one(3), "exp": one(4), "atan2": two(5),
            "asin": one(6), "pow": two(7)}


def vec(values, k="f"):
    return wi.Vec([wi.convert(float(v) if k == "f" else int(v), k) for v in values])


def f32s(v):
    return [float(x) for x in v.e] if isinstance(v, wi.Vec) else float(v)


def uniforms_struct(interp, block):
    return wi.struct_from_record(interp, "Uniforms", block.data[0])


def buffer(interp, name, records):
    return wi.Arr([wi.struct_from_record(interp, name, r) for r in records])


def small_env(rng):
    """A 1024x512 map would be slow to build as Python objects per texel -- the Texture class keeps
    the numpy array, so the full-size synthetic map is used as is."""
    return scenes.synthetic_env()


def hit_row(h):
    return [1.0 if h.f["hit"] else 0.0, float(h.f["t"])] + f32s(h.f["position"]) + f32s(h.f["normal"]) + [float(h.f["materialIndex"])]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "wgsl_vectors.npz")
    shader_dir = os.path.join(ref, "src", "passes", "shaders")
    src = {n: open(os.path.join(shader_dir, n + ".wgsl")).read() for n in ("raytrace", "accumulate", "fullscreen")}
    math_fns = pinned_math()
    rng = np.random.default_rng(20261003)
    out = {}
    t0 = time.time()

    # ------------------------------------------------------------------ raytrace.wgsl
    demo = scenes.demo_scene()
    demo.build_bvh()
    env = small_env(rng)
    rt = wi.Interpreter(src["raytrace"], math_fns)
    rt.res["triangleBuffer"] = buffer(rt, "Triangle", demo.triangles)
    rt.res["materialBuffer"] = buffer(rt, "Material", np.frombuffer(demo.material_bytes.tobytes(), layout.MATERIAL))
    rt.res["bvhBuffer"] = buffer(rt, "BVHNode", demo.nodes)
    rt.res["environmentTexture"] = wi.Texture(env, "linear", "clamp")
    rt.res["environmentTextureSampler"] = "sampler"
    rt.res["environmentCDFTexture"] = wi.Texture(np.zeros((2, 2, 4), F32), "nearest", "clamp")
    rt.res["environmentCDFTextureSampler"] = "sampler"

    def set_uniforms(w, h, frame=2, bounces=3, spf=1, aperture=0.0, focal=1.0, rotation=0.0, intensity=1.0, cam=None, scaling=1.0):
        u = layout.UniformBlock(layout.RAYTRACE_UNIFORMS)
        cam = cam or dict(position=demo.camera["position"], direction=demo.camera_direction(), fov=demo.camera["fov"])
        # raytrace.ts:371-378: resolution = scaledWidth x scaledHeight (may be fractional), aspect = width / height
        u.set({"resolution": [w * scaling, h * scaling], "aspect": w / h, "frame": frame, "maxBounces": bounces, "samplesPerFrame": spf,
               "camera": {"position": cam["position"], "direction": cam["direction"], "fov": cam["fov"],
                          "focalDistance": focal, "aperture": aperture},
               "envMapIntensity": intensity, "envMapRotation": rotation})
        rt.res["uniforms"] = uniforms_struct(rt, u)
        return u

    # rand / sampling helpers
    seeds = np.array([0, 1, 2, 12345, 123456789, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 719393 * 7 + 99], np.uint32)
    rows, tails = [], []
    for s in seeds:
        cell = {"s": wi.U32(int(s))}
        rows.append([float(rt.invoke("rand", [wi.Ref(cell, "s")])) for _ in range(12)])
        tails.append(int(cell["s"]))
    out["rand_seeds"], out["rand_values"], out["rand_final_seed"] = seeds, np.array(rows, F32), np.array(tails, np.uint32)
    for fn, width in (("randPointInCircle", 2), ("randDirection", 3)):
        rows, tails = [], []
        for s in seeds:
            cell = {"s": wi.U32(int(s))}
            rows.append(f32s(rt.invoke(fn, [wi.Ref(cell, "s")])))
            tails.append(int(cell["s"]))
        out[fn + "_values"], out[fn + "_final_seed"] = np.array(rows, F32), np.array(tails, np.uint32)
    normals = rng.normal(size=(len(seeds), 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    rows = []
    for s, n in zip(seeds, normals.astype(F32)):
        cell = {"s": wi.U32(int(s))}
        rows.append(f32s(rt.invoke("randCosineWeightedHemisphere", [wi.Ref(cell, "s"), vec(n)])))
    out["cosine_normals"], out["cosine_values"] = normals.astype(F32), np.array(rows, F32)

    # slab test
    n = 400
    o = (rng.normal(size=(n, 3)) * 2).astype(F32)
    d = rng.normal(size=(n, 3)).astype(F32)
    d[::9, 0] = 0.0
    d[::13, 1] = F32(5e-7)
    d[::17, 2] = F32(-1e-6)
    lo = (rng.uniform(-1.5, 0.5, size=(n, 3))).astype(F32)
    hi = (lo + rng.uniform(0, 2, size=(n, 3)).astype(F32)).astype(F32)
    hi[::7, 1] = lo[::7, 1]                                  # flat boxes
    o[::11] = ((lo[::11] + hi[::11]) * F32(0.5)).astype(F32)  # origins inside
    aim = (lo + (hi - lo) * rng.uniform(-0.15, 1.15, size=(n, 3))).astype(F32)
    sel = np.arange(n) % 3 != 0
    d[sel] = (aim[sel] - o[sel]).astype(F32)                 # two thirds aim at (or just past) the box
    d[::9, 0] = 0.0
    d[::13, 1] = F32(5e-7)
    res = []
    for i in range(n):
        ray = wi.Struct("Ray", {"origin": vec(o[i]), "direction": vec(d[i])})
        res.append(bool(rt.invoke("rayAABBIntersect", [ray, vec(lo[i]), vec(hi[i])])))
    out["aabb_o"], out["aabb_d"], out["aabb_min"], out["aabb_max"], out["aabb_hit"] = o, d, lo, hi, np.array(res, np.uint8)

    # Moller-Trumbore
    n = 300
    tris = np.zeros(n, layout.TRIANGLE)
    for k in ("aPosition", "bPosition", "cPosition"):
        tris[k] = rng.normal(size=(n, 3)).astype(F32)
    for k in ("aNormal", "bNormal", "cNormal"):
        v = rng.normal(size=(n, 3))
        tris[k] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(F32)
    tris["materialIndex"] = rng.integers(0, 5, n)
    o = (rng.normal(size=(n, 3)) * 3).astype(F32)
    target = (tris["aPosition"] + tris["bPosition"] + tris["cPosition"]) / F32(3) + (rng.normal(size=(n, 3)) * 0.4).astype(F32)
    d = (target - o).astype(F32)
    d[::2] = (d[::2] / np.linalg.norm(d[::2], axis=1, keepdims=True)).astype(F32)
    d[5] = (tris["bPosition"][5] - tris["aPosition"][5]).astype(F32)      # in the triangle's plane
    o[7] = (tris["aPosition"][7] + F32(1e-7) * d[7]).astype(F32)          # t below EPSILON
    rows = []
    for i in range(n):
        ray = wi.Struct("Ray", {"origin": vec(o[i]), "direction": vec(d[i])})
        rows.append(hit_row(rt.invoke("rayTriangleIntersect", [ray, wi.struct_from_record(rt, "Triangle", tris[i])])))
    out["tri_records"], out["tri_o"], out["tri_d"], out["tri_hit"] = tris.view(np.uint8).reshape(n, 112).copy(), o, d, np.array(rows, F32)

    # whole traversal on the demo scene
    set_uniforms(8, 8)
    n = 160
    o = (rng.normal(size=(n, 3)) * 3.0).astype(F32)
    t = (rng.normal(size=(n, 3)) * 0.8 + np.array([0.0, 0.4, 0.0])).astype(F32)
    d = t - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F32)
    rows = []
    for i in range(n):
        ray = wi.Struct("Ray", {"origin": vec(o[i]), "direction": vec(d[i])})
        rows.append(hit_row(rt.invoke("raySceneIntersect", [ray])))
    out["scene_o"], out["scene_d"], out["scene_hit"] = o, d, np.array(rows, F32)

    # the 64-entry stack abort (raytrace.wgsl:167-171) on hand-made chain trees, and the strict '<' tie rule
    # on coincident triangles: the scenes are stored with the vectors
    def chain(depth):
        ntri = depth + 1
        pos = np.zeros((ntri, 3, 3))
        for i in range(ntri):
            pos[i] = [[-1, -1, -1.0 - i], [1, -1, -1.0 - i], [0, 1, -1.0 - i]]
        tris = layout.pack_triangles(pos, np.tile(np.array([0.0, 0.0, 1.0]), (ntri, 3, 1)), np.arange(ntri) % 4)
        nodes = np.zeros(2 * ntri - 1, layout.BVH_NODE)
        box = lambda a, b: (pos[a:b + 1].reshape(-1, 3).min(0), pos[a:b + 1].reshape(-1, 3).max(0))
        idx = 0
        for k in range(depth):
            nodes[idx]["min"], nodes[idx]["max"] = box(k, depth)
            nodes[idx]["isLeaf"], nodes[idx]["left"], nodes[idx]["right"], nodes[idx]["triangleIndex"] = 0, idx + 1, idx + 2, -1
            nodes[idx + 1]["min"], nodes[idx + 1]["max"] = box(k, k)
            nodes[idx + 1]["isLeaf"], nodes[idx + 1]["left"], nodes[idx + 1]["right"], nodes[idx + 1]["triangleIndex"] = 1, -1, -1, k
            idx += 2
        nodes[idx]["min"], nodes[idx]["max"] = box(depth, depth)
        nodes[idx]["isLeaf"], nodes[idx]["left"], nodes[idx]["right"], nodes[idx]["triangleIndex"] = 1, -1, -1, depth
        return tris, nodes

    saved = {k: rt.res[k] for k in ("triangleBuffer", "bvhBuffer")}
    for depth in (10, 62, 63, 64, 70):
        tris, nodes = chain(depth)
        rt.res["triangleBuffer"], rt.res["bvhBuffer"] = buffer(rt, "Triangle", tris), buffer(rt, "BVHNode", nodes)
        rows = []
        for ox, oy in ((0.0, 0.0), (0.3, -0.2), (5.0, 5.0)):
            ray = wi.Struct("Ray", {"origin": vec((ox, oy, 5.0)), "direction": vec((0.0, 0.0, -1.0))})
            rows.append([ox, oy] + hit_row(rt.invoke("raySceneIntersect", [ray])))
        out[f"chain{depth}_tris"] = tris.view(np.uint8).reshape(-1, 112).copy()
        out[f"chain{depth}_nodes"] = nodes.view(np.uint8).reshape(-1, 48).copy()
        out[f"chain{depth}_hits"] = np.array(rows, F32)
    # two coincident triangles with different materials under one root: equal t, first visited wins
    pos = np.array([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]] * 2, np.float64)
    for order in (0, 1):
        tris = layout.pack_triangles(pos, np.tile(np.array([0.0, 0.0, 1.0]), (2, 3, 1)), np.array([1 + order, 2 - order]))
        nodes = np.zeros(3, layout.BVH_NODE)
        for k in range(3):
            nodes[k]["min"], nodes[k]["max"] = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
        nodes[0]["isLeaf"], nodes[0]["left"], nodes[0]["right"], nodes[0]["triangleIndex"] = 0, 1, 2, -1
        for k in (1, 2):
            nodes[k]["isLeaf"], nodes[k]["left"], nodes[k]["right"], nodes[k]["triangleIndex"] = 1, -1, -1, k - 1
        rt.res["triangleBuffer"], rt.res["bvhBuffer"] = buffer(rt, "Triangle", tris), buffer(rt, "BVHNode", nodes)
        ray = wi.Struct("Ray", {"origin": vec((0.0, 0.0, 2.0)), "direction": vec((0.0, 0.0, -1.0))})
        out[f"tie{order}_tris"] = tris.view(np.uint8).reshape(-1, 112).copy()
        out[f"tie{order}_nodes"] = nodes.view(np.uint8).reshape(-1, 48).copy()
        out[f"tie{order}_hit"] = np.array(hit_row(rt.invoke("raySceneIntersect", [ray])), F32)
    rt.res.update(saved)

    # camera
    cams = [dict(position=(0.0, 1.0, 4.0), direction=demo.camera_direction(), fov=45.0),
            dict(position=(1.0, 2.0, -3.0), direction=(0.0, -1.0, 0.0), fov=60.0),           # the |w.up| > 0.99999 branch
            dict(position=(-2.0, 0.5, 0.1), direction=(0.6, -0.1, -0.79), fov=23.5)]
    cam_in, cam_out = [], []
    for ci, cam in enumerate(cams):
        for (w, h) in ((64, 64), (1920, 1080)):
            u = set_uniforms(w, h, cam=cam)
            for uv in ((0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.5, 0.5), (0.123, 0.877)):
                ray = rt.invoke("cameraToRay", [rt.res["uniforms"].f["camera"], vec(uv)])
                cam_in.append(list(u.tobytes()) + [0] * 0)
                cam_out.append(list(uv) + f32s(ray.f["origin"]) + f32s(ray.f["direction"]))
    out["camera_uniforms"] = np.array(cam_in, np.uint8)
    out["camera_uv_ray"] = np.array(cam_out, F32)

    # environment lookup
    dirs = rng.normal(size=(60, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.concatenate([dirs, np.eye(3), -np.eye(3), [[0.3, 1.2, -0.4], [1e-8, 0.0, -1.0]]]).astype(F32)
    rows, blocks = [], []
    for rot in (0.0, 0.7, -2.0):
        u = set_uniforms(8, 8, rotation=rot)
        for dd in dirs:
            ray = wi.Struct("Ray", {"origin": vec((0, 0, 0)), "direction": vec(dd)})
            uv = rt.invoke("getEnvironmentMapUVFromRay", [ray])
            col = rt.invoke("getEnvironmentMapColor", [uv])
            blocks.append(list(u.tobytes()))
            rows.append(list(map(float, dd)) + f32s(uv) + f32s(col))
    out["env_uniforms"], out["env_dir_uv_rgb"] = np.array(blocks, np.uint8), np.array(rows, F32)
    print(f"components done in {time.time() - t0:.1f} s, {rt.calls} function calls", flush=True)

    # whole frames: computeMain over every pixel of small images
    frames = [dict(w=24, h=16, frame=2, bounces=3), dict(w=24, h=16, frame=5, bounces=3, aperture=0.05, focal=4.1),
              dict(w=16, h=12, frame=3, bounces=2, spf=2, rotation=0.7, intensity=1.5), dict(w=8, h=8, frame=2, bounces=0),
              dict(w=11, h=7, frame=9, bounces=4),
              dict(w=24, h=16, frame=2, bounces=2, scaling=0.7)]     # scalingFactor < 1: fractional resolution, sub-rectangle
    for fi, cfg in enumerate(frames):
        w, h = cfg["w"], cfg["h"]
        u = set_uniforms(**cfg)
        tex = wi.Texture(np.zeros((h, w, 4), F32))
        rt.res["outputTexture"] = tex
        for y in range(h + 1):                  # one row and column beyond the image: the bounds check
            for x in range(w + 1):
                rt.invoke("computeMain", [vec((x, y, 0), "u")])
        img = np.zeros((h, w, 4), F32)
        sw, sh = int(w * cfg.get("scaling", 1.0)), int(h * cfg.get("scaling", 1.0))
        assert set(tex.stores) == {(x, y) for y in range(sh) for x in range(sw)}
        for (x, y), v in tex.stores.items():
            img[y, x] = v
        out[f"frame{fi}_uniforms"], out[f"frame{fi}_image"] = np.frombuffer(u.tobytes(), np.uint8), img
        print(f"frame {fi} ({w}x{h}) done, {time.time() - t0:.1f} s", flush=True)

    # the same geometry with a polished metal and an emissive material: reflect / mix / emission
    mats = layout.pack_materials([
        dict(color=(0.9, 0.85, 0.8), roughness=0.15, metalness=0.9, specularColor=(0.95, 0.8, 0.6)),
        dict(color=(1.0, 0.05, 0.05), roughness=0.6, metalness=0.3, specularColor=(1.0, 1.0, 1.0),
             emissive=(2.0, 1.0, 0.5), emissiveIntensity=1.5)])
    rt.res["materialBuffer"] = buffer(rt, "Material", np.frombuffer(mats.tobytes(), layout.MATERIAL))
    cfg = dict(w=20, h=14, frame=4, bounces=5)
    u = set_uniforms(**cfg)
    tex = wi.Texture(np.zeros((cfg["h"], cfg["w"], 4), F32))
    rt.res["outputTexture"] = tex
    for y in range(cfg["h"]):
        for x in range(cfg["w"]):
            rt.invoke("computeMain", [vec((x, y, 0), "u")])
    img = np.zeros((cfg["h"], cfg["w"], 4), F32)
    for (x, y), v in tex.stores.items():
        img[y, x] = v
    out["metal_materials"] = np.frombuffer(mats.tobytes(), np.uint8).copy()
    out["metal_uniforms"], out["metal_image"] = np.frombuffer(u.tobytes(), np.uint8), img
    print(f"metal / emissive frame done, {time.time() - t0:.1f} s", flush=True)

    # the dormant environment importance sampling (raytrace.wgsl:315-367): its call sites are
    # commented out in the shipped shader (:398, :402-404); the three markers are removed from the
    # in-memory text, nothing else is touched
    dormant = src["raytrace"]
    for marker, live in (("// uv = getEnvironmentMapUV(seed);", "uv = getEnvironmentMapUV(seed);"),
                         ("// let pdf = getEnvironmentMapPDF(uv);", "let pdf = getEnvironmentMapPDF(uv);"),
                         ("// incomingLight /= pdf;", "incomingLight /= pdf;")):
        assert dormant.count(marker) == 1, marker
        dormant = dormant.replace(marker, live)
    from mi3pt_host import capi
    cdf = capi.host_env_cdf(env)
    rt2 = wi.Interpreter(dormant, math_fns, dict(rt.res))
    rt2.res["materialBuffer"] = buffer(rt2, "Material", np.frombuffer(demo.material_bytes.tobytes(), layout.MATERIAL))
    rt2.res["environmentCDFTexture"] = wi.Texture(cdf, "nearest", "clamp")
    cfg = dict(w=18, h=12, frame=6, bounces=3)
    u = set_uniforms(**cfg)
    rt2.res["uniforms"] = rt.res["uniforms"]
    tex = wi.Texture(np.zeros((cfg["h"], cfg["w"], 4), F32))
    rt2.res["outputTexture"] = tex
    for y in range(cfg["h"]):
        for x in range(cfg["w"]):
            rt2.invoke("computeMain", [vec((x, y, 0), "u")])
    img = np.zeros((cfg["h"], cfg["w"], 4), F32)
    for (x, y), v in tex.stores.items():
        img[y, x] = v
    out["envsample_uniforms"], out["envsample_image"] = np.frombuffer(u.tobytes(), np.uint8), img
    rows = []
    for s in seeds:
        cell = {"s": wi.U32(int(s))}
        uv = rt2.invoke("getEnvironmentMapUV", [wi.Ref(cell, "s")])
        rows.append(f32s(uv) + [float(rt2.invoke("getEnvironmentMapPDF", [uv])), float(int(cell["s"]) & 0xFFFFFF)])
    out["envsample_uv_pdf"] = np.array(rows, F32)
    print(f"dormant importance-sampling frame done, {time.time() - t0:.1f} s", flush=True)

    # ------------------------------------------------------------------ accumulate.wgsl
    acc = wi.Interpreter(src["accumulate"], math_fns)
    w, h = 9, 6
    cur = rng.uniform(0, 4, size=(h, w, 4)).astype(F32)
    prev = rng.uniform(0, 4, size=(h, w, 4)).astype(F32)
    out["acc_cur"], out["acc_prev"] = cur, prev
    cases = [(0, 1), (1, 1), (2, 1), (7, 1), (1000, 1), (5, 0), (0, 0)]
    res = []
    for frame, enabled in cases:
        u = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS).set({"resolution": [w - 1, h - 1], "frame": frame, "enabled": enabled})
        acc.res["uniforms"] = uniforms_struct(acc, u)
        acc.res["inputTexture"], acc.res["outputTexturePrev"] = wi.Texture(cur), wi.Texture(prev)
        tex = wi.Texture(np.zeros((h, w, 4), F32))
        acc.res["outputTexture"] = tex
        for y in range(h):
            for x in range(w):
                acc.invoke("computeMain", [vec((x, y, 0), "u")])
        img = np.full((h, w, 4), np.nan, F32)                  # untouched texels (outside resolution) stay NaN
        for (x, y), v in tex.stores.items():
            img[y, x] = v
        res.append(img)
    out["acc_cases"], out["acc_out"] = np.array(cases, np.uint32), np.array(res, F32)

    # ------------------------------------------------------------------ fullscreen.wgsl
    fs = wi.Interpreter(src["fullscreen"], math_fns)
    w, h = 14, 10
    tex_in = (rng.uniform(0, 1, size=(h, w, 4)) ** 3 * 3).astype(F32)
    tex_in[..., 3] = 1.0
    tex_in[3, 4, :3] = (40.0, 25.0, 3.0)                     # a firefly
    out["fs_input"] = tex_in
    fs.res["inputTexture"] = wi.Texture(tex_in, "linear", "repeat")
    fs.res["inputTextureSampler"] = "sampler"
    cases = [(1, 1, 1.0), (0, 1, 1.0), (1, 0, 1.0), (1, 2, 1.0), (1, 1, 0.5), (0, 2, 0.25)]
    res = []
    for denoise, tonemap, scaling in cases:
        u = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS).set(
            {"resolution": [w, h], "aspect": w / h, "scalingFactor": scaling, "denoise": denoise, "tonemapping": tonemap})
        fs.res["uniforms"] = uniforms_struct(fs, u)
        img = np.zeros((h, w, 4), F32)
        for py in range(h):
            for px in range(w):
                # the quad's uv (0,0) sits at clip (-1,-1) = bottom left; framebuffer row 0 is the top
                uu = F32(F32(F32(px) + F32(0.5)) / F32(w)) * F32(scaling)
                vv = F32(F32(1.0) - F32(F32(py) + F32(0.5)) / F32(h)) * F32(scaling)
                inp = wi.Struct("VertexOutput", {"position": vec((0, 0, 0, 1)), "uv": wi.Vec([F32(uu), F32(vv)])})
                img[py, px] = f32s(fs.invoke("fragmentMain", [inp]))
        res.append(img)
        print(f"fullscreen case {denoise, tonemap, scaling} done, {time.time() - t0:.1f} s", flush=True)
    out["fs_cases"], out["fs_out"] = np.array(cases, F32), np.array(res, F32)

    np.savez_compressed(out_path, **out)
    print(f"wrote {out_path}: {len(out)} arrays, {os.path.getsize(out_path)} bytes, {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
