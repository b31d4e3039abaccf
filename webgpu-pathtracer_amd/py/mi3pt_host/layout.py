"""Byte layouts of the reference's GPU structs, as numpy structured dtypes.

These are the offsets webgpu-utils computes from the WGSL declarations (reference:
src/passes/raytrace.ts:53,89-94,123-128,162-167,195-197):

    Triangle  112 B  raytrace.wgsl:40-49      BVHNode  48 B  raytrace.wgsl:51-64
    Material   64 B  raytrace.wgsl:31-38      Uniforms 96 B  raytrace.wgsl:66-75
    accumulate Uniforms 16 B accumulate.wgsl:1-5
    fullscreen Uniforms 24 B fullscreen.wgsl:14-20

`UniformBlock` mirrors webgpu-utils' StructuredView.set(): a *partial* set that leaves
unnamed fields untouched (renderer.ts:358-360 relies on that).
"""
import numpy as np

TRIANGLE = np.dtype({
    "names": ["aPosition", "bPosition", "cPosition", "aNormal", "bNormal", "cNormal",
              "materialIndex", "aabbIndex"],
    "formats": [("<f4", 3)] * 6 + ["<i4", "<i4"],
    "offsets": [0, 16, 32, 48, 64, 80, 92, 96],
    "itemsize": 112,
})

BVH_NODE = np.dtype({
    "names": ["min", "max", "isLeaf", "left", "right", "triangleIndex"],
    "formats": [("<f4", 3), ("<f4", 3), "<i4", "<i4", "<i4", "<i4"],
    "offsets": [0, 16, 28, 32, 36, 40],
    "itemsize": 48,
})

MATERIAL = np.dtype({
    "names": ["color", "specularColor", "roughness", "metalness", "emissionColor", "emissionStrength"],
    "formats": [("<f4", 3), ("<f4", 3), "<f4", "<f4", ("<f4", 3), "<f4"],
    "offsets": [0, 16, 28, 32, 48, 60],
    "itemsize": 64,
})

RAYTRACE_UNIFORMS = np.dtype({
    "names": ["resolution", "aspect", "frame", "maxBounces", "samplesPerFrame",
              "camera.position", "camera.direction", "camera.fov", "camera.focalDistance",
              "camera.aperture", "envMapIntensity", "envMapRotation"],
    "formats": [("<f4", 2), "<f4", "<u4", "<i4", "<i4", ("<f4", 3), ("<f4", 3), "<f4", "<f4", "<f4",
                "<f4", "<f4"],
    "offsets": [0, 8, 12, 16, 20, 32, 48, 60, 64, 68, 80, 84],
    "itemsize": 96,
})

ACCUMULATE_UNIFORMS = np.dtype({
    "names": ["resolution", "frame", "enabled"],
    "formats": [("<u4", 2), "<u4", "<u4"],
    "offsets": [0, 8, 12],
    "itemsize": 16,
})

FULLSCREEN_UNIFORMS = np.dtype({
    "names": ["resolution", "aspect", "scalingFactor", "denoise", "tonemapping"],
    "formats": [("<f4", 2), "<f4", "<f4", "<u4", "<u4"],
    "offsets": [0, 8, 12, 16, 20],
    "itemsize": 24,
})

ENV_WIDTH, ENV_HEIGHT = 1024, 512


def _flatten(value, prefix=""):
    for key, val in value.items():
        name = prefix + key
        if isinstance(val, dict):
            yield from _flatten(val, name + ".")
        else:
            yield name, val


class UniformBlock:
    """One uniform struct with webgpu-utils' partial `set` semantics."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.data = np.zeros(1, dtype)

    def set(self, value):
        for name, val in _flatten(value):
            if name not in self.dtype.names:
                continue                      # webgpu-utils ignores unknown keys
            field = self.dtype.fields[name][0]
            if field.kind == "u":
                # typed-array store of a JS number: ToUint32 (truncation toward zero)
                arr = np.trunc(np.asarray(val, np.float64)).astype(np.int64) & 0xFFFFFFFF
                self.data[name] = arr.astype(np.uint32)
            elif field.kind == "i" and field.shape == ():
                self.data[name] = np.int32(np.trunc(val))
            else:
                self.data[name] = val
        return self

    def get(self, name):
        return self.data[name][0]

    def tobytes(self):
        return self.data.tobytes()


def pack_materials(materials):
    """raytrace.ts:138-160 -- `materials` is a list of dicts with the reference's
    RaytracingMaterial fields (color, specularColor, roughness, metalness, emissive,
    emissiveIntensity)."""
    out = np.zeros(len(materials), MATERIAL)
    for i, m in enumerate(materials):
        out[i]["color"] = m.get("color", (1.0, 1.0, 1.0))
        out[i]["specularColor"] = m.get("specularColor", (1.0, 1.0, 1.0))
        out[i]["roughness"] = m.get("roughness", 1.0)
        out[i]["metalness"] = m.get("metalness", 0.0)
        out[i]["emissionColor"] = m.get("emissive", (0.0, 0.0, 0.0))
        out[i]["emissionStrength"] = m.get("emissiveIntensity", 1.0)
    return out


def pack_triangles(positions, normals, material_index):
    """raytrace.ts:104-121 -- positions/normals: (n, 3, 3) float64 world-space vertex
    data (a, b, c); material_index: (n,) ints.  Stores round to fp32 like the view."""
    n = len(positions)
    out = np.zeros(n, TRIANGLE)
    p = np.asarray(positions, np.float64).astype(np.float32)
    nn = np.asarray(normals, np.float64).astype(np.float32)
    out["aPosition"], out["bPosition"], out["cPosition"] = p[:, 0], p[:, 1], p[:, 2]
    out["aNormal"], out["bNormal"], out["cNormal"] = nn[:, 0], nn[:, 1], nn[:, 2]
    out["materialIndex"] = np.asarray(material_index, np.int32)
    return out
