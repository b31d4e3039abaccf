"""Shared helpers for the tests: uniform blocks, scene upload, image comparison."""
import numpy as np

from mi3pt_host import capi, layout


def rt_uniforms(sc, w, h, frame=2, bounces=4, spf=1, aperture=None, focal=None, res=None,
                intensity=1.0, rotation=0.0, aspect=None, fov=None, position=None, direction=None):
    cam = sc.camera
    u = layout.UniformBlock(layout.RAYTRACE_UNIFORMS)
    u.set({"resolution": list(res) if res is not None else [w, h],
           "aspect": aspect if aspect is not None else w / h,
           "frame": frame, "maxBounces": bounces, "samplesPerFrame": spf,
           "camera": {"position": position if position is not None else cam["position"],
                      "direction": direction if direction is not None else sc.camera_direction(),
                      "fov": fov if fov is not None else cam["fov"],
                      "focalDistance": focal if focal is not None else cam["focalDistance"],
                      "aperture": aperture if aperture is not None else cam["aperture"]},
           "envMapIntensity": intensity, "envMapRotation": rotation})
    return u


def acc_uniforms(w, h, frame, enabled=1):
    u = layout.UniformBlock(layout.ACCUMULATE_UNIFORMS)
    u.set({"resolution": [w, h], "frame": frame, "enabled": enabled})
    return u


def fs_uniforms(w, h, scaling=1.0, denoise=1, tonemapping=1):
    u = layout.UniformBlock(layout.FULLSCREEN_UNIFORMS)
    u.set({"resolution": [w, h], "aspect": w / h, "scalingFactor": scaling, "denoise": denoise,
           "tonemapping": tonemapping})
    return u


def upload_scene(ctx, sc, env=None):
    ctx.upload_bvh(sc.nodes)
    ctx.upload_triangles(sc.triangles)
    ctx.upload_materials(sc.material_bytes)
    if env is not None:
        ctx.upload_environment(env)


def oracle_scene(orc, sc, env=None):
    return orc.OracleScene(sc.triangles, sc.material_bytes, sc.nodes, env)


def same_bits(a, b):
    """Bit-for-bit equality of two float arrays (NaNs must match as NaNs)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    eq = (a == b) | (np.isnan(a) & np.isnan(b))
    return bool(eq.all())


def describe_diff(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    n = int(bad.sum())
    if n == 0:
        return "identical"
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-12)
    idx = np.argwhere(bad)[:5].tolist()
    return f"{n} of {a.size} values differ; max rel {np.nanmax(rel[bad]):.3g}; first at {idx}"


def max_rel_err(a, b, floor=1e-6):
    """Per-value relative error |a-b| / max(|b|, floor), NaN-aware (NaN==NaN is 0)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    rel = np.abs(a - b) / np.maximum(np.abs(b), floor)
    rel[both_nan] = 0.0
    return float(np.nanmax(rel)) if rel.size else 0.0


def gpu_frame(ctx, rt_u, acc_u=None, mask=capi.SUBMIT_RAYTRACE):
    ctx.set_uniforms(capi.PASS_RAYTRACE, rt_u.tobytes())
    if acc_u is not None:
        ctx.set_uniforms(capi.PASS_ACCUMULATE, acc_u.tobytes())
    ctx.submit(mask)


PATH_COUNTERS = ("rays", "hits", "misses", "stack_overflows", "pixels")
WALK_COUNTERS = ("box_tests", "tri_tests")


def check_counters(cnt, want, culled=False, what=""):
    """Counters of a raytrace run against the oracle's (or a reference kernel's).  Kernel variants
    1-8 execute exactly the reference's tests, so everything is equal.  The distance-culling walks
    (variants 9 and 10, the default when the scene allows it) trace the same paths -- rays, hits,
    misses, pixels equal -- and never test a triangle the reference does not test.  Box tests:
    variant 9 skips boxes behind the closest hit (at or below the reference's count); variant 10
    walks 4-ary packets, i.e. it tests the four grandchildren where the reference tests two
    children and then, maybe, theirs -- fewer node steps, not necessarily fewer box tests (a ray
    that hits nothing can test more), so the count is not compared."""
    for k in PATH_COUNTERS:
        if k in want:
            assert cnt[k] == want[k], f"{what} counter {k}: gpu {cnt[k]} reference {want[k]}"
    if culled:
        assert cnt["tri_tests"] <= want["tri_tests"], \
            f"{what} counter tri_tests: gpu {cnt['tri_tests']} above the reference walk's {want['tri_tests']}"
    else:
        for k in WALK_COUNTERS:
            assert cnt[k] == want[k], f"{what} counter {k}: gpu {cnt[k]} reference {want[k]}"


def set_variant_or_skip(ctx, variant):
    """Kernel variants 3, 5, 6, 8 (measured, not adopted) and 11, 12 (superseded by 13) only exist in the experiment build of the library
    (`make -C webgpu-pathtracer_amd/csrc experiments`, loaded through MI3PT_LIBRARY): a release library refuses them."""
    import pytest
    from mi3pt_host import capi
    try:
        ctx.set_kernel_variant(variant)
    except capi.Mi3ptError as e:
        if "experiment build only" in e.message:
            pytest.skip(f"kernel variant {variant}: experiment build only")
        raise


def variants_available(ctx, variants):
    """The subset of `variants` this build of the library has (see set_variant_or_skip)."""
    from mi3pt_host import capi
    out = []
    for v in variants:
        try:
            ctx.set_kernel_variant(v)
            out.append(v)
        except capi.Mi3ptError as e:
            if "experiment build only" not in e.message:
                raise
    ctx.set_kernel_variant(0)
    return tuple(out)
