"""Known-answer tests for oracle/wgsl_interp.py, the interpreter that executes the reference's shader TEXT and so pins
the oracle and the kernels (tests/golden/make_wgsl_vectors.py, tests/test_wgsl_vectors.py).  Everything else in the
repository is checked against what this interpreter computes; these cases check the interpreter against the WGSL
specification itself (W3C "WebGPU Shading Language", section titles quoted per case) -- expected values are written
down from the specification's rules and IEEE-754 arithmetic (numpy / Python integers), never from the oracle, the
kernels or the shaders.  The constructs are the ones the three shader files use; the reference call sites that
depend on each rule are cited (src/passes/shaders/*.wgsl)."""
import struct

import numpy as np
import pytest

from oracle import wgsl_interp as wi

F32 = np.float32
MATH = {"sin": np.sin, "cos": np.cos, "tan": np.tan, "log": np.log, "exp": np.exp, "asin": np.arcsin,
        "atan2": np.arctan2, "pow": np.power}


def run(src, fn="t", args=(), resources=None):
    return wi.Interpreter(src, MATH, resources).invoke(fn, list(args))


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def f(x):
    return wi.F32(x)


def u(x):
    return wi.convert(int(x), "u")


def i(x):
    return wi.convert(int(x), "i")


def vals(v):
    return [float(x) for x in v.e]


# ------------------------------------------------------------------ abstract numerics and conversions

def test_abstract_float_const_expression_is_rounded_once():
    """Spec "Abstract Numeric Types" + "Conversion Rank": a const-expression of AbstractFloat literals is evaluated as
    AbstractFloat (binary64) and converted to f32 once.  fullscreen.wgsl:99 `vec3f(1.0 / 2.2)`: the double quotient
    rounded to f32 is 0x3EE8BA2F; dividing the two f32 values gives 0x3EE8BA2E (the transcription error this found)."""
    r = run("fn t() -> f32 { let g : f32 = 1.0 / 2.2; return g; }")
    assert bits(r) == 0x3EE8BA2F == bits(np.float32(1.0 / 2.2))
    assert bits(np.float32(1.0) / np.float32(2.2)) == 0x3EE8BA2E
    r = run("fn t() -> f32 { let a : f32 = 1.0; let b : f32 = 2.2; return a / b; }")
    assert bits(r) == 0x3EE8BA2E


def test_abstract_float_literal_to_f32_rounds_to_nearest():
    """Spec "Floating Point Conversion": an AbstractFloat converts to f32 by rounding.  raytrace.wgsl:258 divides by
    the literal 4294967295.0, which is not representable: it becomes 2^32."""
    assert float(run("fn t() -> f32 { let a : f32 = 4294967295.0; return a; }")) == 4294967296.0
    assert float(run("fn t() -> f32 { return f32(1) / 4294967295.0; }")) == 2.0 ** -32


def test_let_without_type_concretises_abstract_values():
    """Spec "Value Declarations": `let x = 1;` is i32, `let y = 1.5;` is f32, `let z = 1u;` is u32."""
    it = wi.Interpreter("fn a() -> i32 { let x = 7; return x; } fn b() -> f32 { let y = 1.5; return y; } fn c() -> u32 { let z = 3u; return z; }", MATH)
    assert wi.kind(it.invoke("a", [])) == "i" and wi.kind(it.invoke("b", [])) == "f" and wi.kind(it.invoke("c", [])) == "u"


def test_abstract_int_meets_f32_in_a_mixed_expression():
    """Spec "Conversion Rank": AbstractInt converts to f32 when the other operand is f32 (raytrace.wgsl:402 `2 * ...`)."""
    r = run("fn t(x : f32) -> f32 { return 2 * x + 1; }", args=[f(0.25)])
    assert wi.kind(r) == "f" and float(r) == 1.5


def test_abstract_int_is_64_bit_until_converted():
    """Spec "Abstract Numeric Types": AbstractInt is a 64-bit integer; 4294967295 fits it and f32() rounds it."""
    assert float(run("const BIG = 4294967295; fn t() -> f32 { return f32(BIG); }")) == 4294967296.0


@pytest.mark.parametrize("value,want", [(16777217, 16777216.0), (16777219, 16777220.0), (4294967295, 4294967296.0), (0, 0.0), (123456789, float(np.float32(123456789)))])
def test_f32_of_u32_rounds_to_nearest_even(value, want):
    """Spec "Floating Point Conversion" (integer to floating point: nearest, ties to even).  raytrace.wgsl:258 f32(result)."""
    assert float(run("fn t(x : u32) -> f32 { return f32(x); }", args=[u(value)])) == want


@pytest.mark.parametrize("value,want", [(3.99, 3), (0.5, 0), (-0.5, 0), (-7.0, 0), (4294967296.0, 4294967295), (1e20, 4294967295), (719393.0, 719393)])
def test_u32_of_f32_clamps_then_truncates(value, want):
    """Spec "Floating Point Conversion": floating point to integer -- clamped to the target range, then rounded toward
    zero.  raytrace.wgsl:431-436 `u32(uniforms.resolution.x)`; accumulate.ts writes floats that the shader reads as u32."""
    assert int(run("fn t(x : f32) -> u32 { return u32(x); }", args=[f(value)])) == want


@pytest.mark.parametrize("value,want", [(-2.7, -2), (2.7, 2), (3e9, 2147483647), (-3e9, -2147483648), (-0.0, 0)])
def test_i32_of_f32_clamps_then_truncates(value, want):
    """Spec "Floating Point Conversion".  fullscreen.wgsl:69-71 loop bounds are floats; i32() appears in index math."""
    assert int(run("fn t(x : f32) -> i32 { return i32(x); }", args=[f(value)])) == want


def test_bool_and_integer_constructors():
    """Spec "Value Constructor Built-in Functions": u32(i32) reinterprets modulo 2^32, f32(bool) is 0 or 1."""
    assert int(run("fn t(x : i32) -> u32 { return u32(x); }", args=[i(-1)])) == 0xFFFFFFFF
    assert int(run("fn t(x : u32) -> i32 { return i32(x); }", args=[u(0x80000000)])) == -2147483648
    assert float(run("fn t(x : bool) -> f32 { return f32(x); }", args=[True])) == 1.0


# ------------------------------------------------------------------ integer arithmetic

def test_u32_arithmetic_wraps_modulo_2_32():
    """Spec "Arithmetic Expressions" (integer overflow wraps).  The PCG step of raytrace.wgsl:254-256."""
    s = 123456789
    want = (s * 747796405 + 2891336453) & 0xFFFFFFFF
    assert int(run("fn t(s : u32) -> u32 { return s * 747796405u + 2891336453u; }", args=[u(s)])) == want
    assert int(run("fn t(s : u32) -> u32 { return s + 1u; }", args=[u(0xFFFFFFFF)])) == 0
    assert int(run("fn t(s : u32) -> u32 { return s - 1u; }", args=[u(0)])) == 0xFFFFFFFF


def test_i32_arithmetic_wraps():
    """Spec "Arithmetic Expressions"."""
    assert int(run("fn t(s : i32) -> i32 { return s + 1; }", args=[i(2147483647)])) == -2147483648
    assert int(run("fn t(s : i32) -> i32 { return -s; }", args=[i(-2147483648)])) == -2147483648


def test_pcg_output_permutation_matches_the_published_algorithm():
    """Spec "Bit Expressions".  raytrace.wgsl:253-259 is PCG-RXS-M-XS-32 (O'Neill); its output for a state is a
    function of integer rules only -- computed here with Python integers."""
    def pcg(state):
        state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return state, ((word >> 22) ^ word) & 0xFFFFFFFF
    src = """fn t(seed : ptr<function, u32>) -> u32 {
        *seed = *seed * 747796405u + 2891336453u;
        var result : u32 = ((*seed >> ((*seed >> 28u) + 4u)) ^ *seed) * 277803737u;
        result = (result >> 22u) ^ result;
        return result; }
    fn go(s0 : u32) -> vec2u { var s = s0; let a = t(&s); let b = t(&s); return vec2u(a, b); }"""
    for s0 in (0, 1, 123456789, 0xFFFFFFFF, 0x9E3779B9):
        st, a = pcg(s0)
        st, b = pcg(st)
        got = run(src, "go", [u(s0)])
        assert [int(x) for x in got.e] == [a, b]


@pytest.mark.parametrize("x,s,want", [(0x80000000, 31, 1), (1, 31, 0), (0xDEADBEEF, 0, 0xDEADBEEF), (0xDEADBEEF, 32, 0xDEADBEEF), (0xDEADBEEF, 36, 0x0DEADBEE)])
def test_u32_shift_right_is_logical_and_the_count_is_taken_modulo_32(x, s, want):
    """Spec "Bit Expressions": e1 >> e2 on u32 is a logical shift by e2 modulo the bit width (a run-time count of 32
    shifts by 0).  raytrace.wgsl:256 shifts by (seed >> 28u) + 4u, i.e. 4..19."""
    assert int(run("fn t(x : u32, s : u32) -> u32 { return x >> s; }", args=[u(x), u(s)])) == want


def test_shift_left_and_arithmetic_shift_right():
    """Spec "Bit Expressions": << discards overflowing bits; >> on i32 is arithmetic."""
    assert int(run("fn t(x : u32, s : u32) -> u32 { return x << s; }", args=[u(0xC0000001), u(1)])) == 0x80000002
    assert int(run("fn t(x : i32, s : u32) -> i32 { return x >> s; }", args=[i(-8), u(1)])) == -4
    assert int(run("fn t(x : u32, s : u32) -> u32 { return x << s; }", args=[u(3), u(33)])) == 6


def test_bitwise_operators():
    """Spec "Bit Expressions"."""
    assert int(run("fn t(a : u32, b : u32) -> u32 { return (a ^ b) | (a & b); }", args=[u(0b1100), u(0b1010)])) == 0b1110


@pytest.mark.parametrize("a,b,q,r", [(7, 2, 3, 1), (-7, 2, -3, -1), (7, -2, -3, 1), (-7, -2, 3, -1)])
def test_integer_division_truncates_and_remainder_follows_the_dividend(a, b, q, r):
    """Spec "Arithmetic Expressions": e1 / e2 truncates toward zero, e1 % e2 = e1 - e2 * trunc(e1 / e2)."""
    got = run("fn t(a : i32, b : i32) -> vec2i { return vec2i(a / b, a % b); }", args=[i(a), i(b)])
    assert [int(x) for x in got.e] == [q, r]


def test_u32_division_and_index_arithmetic():
    """Spec "Arithmetic Expressions".  raytrace.wgsl:435 `id.x + id.y * u32(resolution.x)`."""
    assert int(run("fn t(x : u32, y : u32, w : f32) -> u32 { return x + y * u32(w); }", args=[u(5), u(3), f(1920.0)])) == 5765
    assert int(run("fn t(a : u32) -> u32 { return a / 2u; }", args=[u(7)])) == 3


# ------------------------------------------------------------------ floating point

def test_f32_operations_are_correctly_rounded_binary32():
    """Spec "Floating Point Evaluation" + IEEE-754: + - * / and sqrt on f32 (the pinned interpretation: correctly rounded)."""
    assert bits(run("fn t(a : f32, b : f32) -> f32 { return a / b; }", args=[f(1.0), f(3.0)])) == 0x3EAAAAAB
    assert bits(run("fn t(a : f32) -> f32 { return sqrt(a); }", args=[f(2.0)])) == 0x3FB504F3
    assert bits(run("fn t(a : f32, b : f32) -> f32 { return a * b + a; }", args=[f(1.0 + 2.0 ** -23), f(1.0 + 2.0 ** -23)])) == \
        bits(np.float32(np.float32(np.float32(1.0 + 2.0 ** -23) * np.float32(1.0 + 2.0 ** -23)) + np.float32(1.0 + 2.0 ** -23)))     # two roundings: no contraction
    assert bits(run("fn t(a : f32, b : f32) -> f32 { return a + b; }", args=[f(16777216.0), f(1.0)])) == bits(16777216.0)


def test_float_remainder_is_truncated():
    """Spec "Arithmetic Expressions": e1 % e2 = e1 - e2 * trunc(e1 / e2)."""
    assert float(run("fn t(a : f32, b : f32) -> f32 { return a % b; }", args=[f(5.5), f(2.0)])) == 1.5
    assert float(run("fn t(a : f32, b : f32) -> f32 { return a % b; }", args=[f(-5.5), f(2.0)])) == -1.5


def test_float_loop_terminates_at_the_denoise_bounds():
    """Spec "For Statement" + f32 arithmetic: fullscreen.wgsl:69-71 `for (var x = -5.0; x <= 5.0; x = x + 1.0)` runs
    11 times per axis (121 taps); the counter is f32."""
    src = """fn t() -> vec2f { var n = 0.0; var last = 0.0;
        for (var x = -5.0; x <= 5.0; x = x + 1.0) { for (var y = -5.0; y <= 5.0; y = y + 1.0) { n = n + 1.0; last = y; } }
        return vec2f(n, last); }"""
    assert vals(run(src)) == [121.0, 5.0]


# ------------------------------------------------------------------ statements, pointers, composite values

def test_compound_assignment_through_a_function_pointer():
    """Spec "Compound Assignment" + "Pointer Types": *p op= e reads and writes the pointee once; the pointee is the
    caller's variable.  raytrace.wgsl:253-259 `fn rand(seed: ptr<function, u32>)` called as rand(&seed)."""
    src = """fn bump(p : ptr<function, u32>) { *p += 2u; *p *= 3u; }
    fn t() -> u32 { var s = 5u; bump(&s); bump(&s); return s; }"""
    assert int(run(src)) == ((5 + 2) * 3 + 2) * 3


def test_increment_and_decrement_statements():
    """Spec "Increment and Decrement Statements".  raytrace.wgsl:177 `stackPointer--`."""
    assert int(run("fn t() -> i32 { var a = 3; a++; a++; a--; return a; }")) == 4


def test_variables_have_value_semantics():
    """Spec "Variable and Value Declarations": `var b = a;` copies; writing b leaves a alone (structs and vectors)."""
    src = """struct S { v : vec3f, k : u32 }
    fn t() -> vec4f { var a = S(vec3f(1.0, 2.0, 3.0), 7u); var b = a; b.v.x = 9.0; b.k = 1u; return vec4f(a.v.x, b.v.x, f32(a.k), f32(b.k)); }"""
    assert vals(run(src)) == [1.0, 9.0, 7.0, 1.0]


def test_function_arguments_are_passed_by_value():
    """Spec "Function Calls"."""
    src = "fn g(v : vec3f) -> f32 { var w = v; w.x = 100.0; return w.x; } fn t() -> vec2f { let a = vec3f(1.0); let r = g(a); return vec2f(a.x, r); }"
    assert vals(run(src)) == [1.0, 100.0]


def test_single_component_assignment_and_swizzle_reads():
    """Spec "Vector Access Expression": one component may be assigned; a multi-letter swizzle is a value (read only)."""
    assert vals(run("fn t() -> vec4f { var v = vec4f(1.0, 2.0, 3.0, 4.0); v.z = 9.0; let s = v.wzyx; return s; }")) == [4.0, 9.0, 2.0, 1.0]
    assert vals(run("fn t() -> vec2f { let v = vec4f(1.0, 2.0, 3.0, 4.0); return v.rg + v.ba; }")) == [4.0, 6.0]


def test_vector_constructors():
    """Spec "Value Constructor Built-in Functions": splat, concatenation, component conversion."""
    assert vals(run("fn t() -> vec3f { return vec3f(0.5); }")) == [0.5, 0.5, 0.5]
    assert vals(run("fn t() -> vec4f { let a = vec3f(1.0, 2.0, 3.0); return vec4f(a, 1.0); }")) == [1.0, 2.0, 3.0, 1.0]
    assert vals(run("fn t() -> vec3f { let a = vec2f(1.0, 2.0); return vec3f(a.yx, 7.0); }")) == [2.0, 1.0, 7.0]
    assert [int(x) for x in run("fn t() -> vec2u { return vec2u(vec2f(3.7, 9.2)); }").e] == [3, 9]


def test_matrix_is_column_major():
    """Spec "Matrix Types" + "Arithmetic Expressions": mat3x3f(c0, c1, c2) has COLUMNS c0 c1 c2; m * v = v.x c0 + v.y c1 + v.z c2;
    m[i] is column i.  fullscreen.wgsl:88-97 ACES input / output matrices."""
    src = """fn t(v : vec3f) -> vec3f { let m = mat3x3f(vec3f(1.0, 2.0, 3.0), vec3f(10.0, 20.0, 30.0), vec3f(100.0, 200.0, 300.0)); return m * v; }"""
    assert vals(run(src, args=[wi.Vec([f(1.0), f(0.0), f(0.0)])])) == [1.0, 2.0, 3.0]
    assert vals(run(src, args=[wi.Vec([f(1.0), f(2.0), f(3.0)])])) == [321.0, 642.0, 963.0]


def test_arrays_and_structs():
    """Spec "Array Types", "Structure Types", arrayLength.  raytrace.wgsl:156 `var stack: array<i32, MAX_STACK_SIZE>`."""
    src = """const N = 4;
    fn t() -> i32 { var a : array<i32, N>; a[0] = 5; a[3] = a[0] + 2; var s = 0; for (var k = 0; k < N; k++) { s += a[k]; } return s; }"""
    assert int(run(src)) == 12


def test_control_flow():
    """Spec "Control Flow": break / continue / while / early return, && and || short-circuit."""
    src = """fn side(c : ptr<function, i32>) -> bool { *c += 1; return true; }
    fn t() -> vec3i { var n = 0; var calls = 0;
        for (var k = 0; k < 10; k++) { if (k == 2) { continue; } if (k == 5) { break; } n += k; }
        var w = 0; while (w < 3) { w++; }
        let a = false && side(&calls); let b = true || side(&calls); let c = true && side(&calls);
        return vec3i(n, w, calls); }"""
    assert [int(x) for x in run(src).e] == [0 + 1 + 3 + 4, 3, 1]


def test_comparisons_and_select():
    """Spec "Comparison Expressions", select(f, t, cond) returns t when cond is true."""
    assert float(run("fn t(x : f32) -> f32 { return select(1.0, 2.0, x > 0.5); }", args=[f(0.75)])) == 2.0
    assert float(run("fn t(x : f32) -> f32 { return select(1.0, 2.0, x != x); }", args=[f(0.75)])) == 1.0


# ------------------------------------------------------------------ numeric built-ins (spec "Numeric Built-in Functions")

def test_dot_cross_length_normalize_reflect_are_the_spec_formulas():
    """dot = sum of products; cross per the spec's component formula; length = sqrt(dot(e, e)); normalize = e / length(e);
    reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2."""
    a, b = wi.Vec([f(1.0), f(2.0), f(3.0)]), wi.Vec([f(-4.0), f(0.5), f(2.0)])
    assert float(run("fn t(a : vec3f, b : vec3f) -> f32 { return dot(a, b); }", args=[a, b])) == -4.0 + 1.0 + 6.0
    assert vals(run("fn t(a : vec3f, b : vec3f) -> vec3f { return cross(a, b); }", args=[a, b])) == [2.0 * 2.0 - 3.0 * 0.5, 3.0 * -4.0 - 1.0 * 2.0, 1.0 * 0.5 - 2.0 * -4.0]
    assert float(run("fn t(a : vec3f) -> f32 { return length(a); }", args=[wi.Vec([f(3.0), f(0.0), f(4.0)])])) == 5.0
    assert vals(run("fn t(a : vec3f) -> vec3f { return normalize(a); }", args=[wi.Vec([f(0.0), f(0.0), f(-8.0)])])) == [0.0, 0.0, -1.0]
    got = vals(run("fn t(i : vec3f, n : vec3f) -> vec3f { return reflect(i, n); }", args=[wi.Vec([f(1.0), f(-1.0), f(0.0)]), wi.Vec([f(0.0), f(1.0), f(0.0)])]))
    assert got == [1.0, 1.0, 0.0]


def test_mix_clamp_min_max_abs_sign_floor():
    """mix(e1, e2, e3) = e1 * (1 - e3) + e2 * e3; clamp(e, lo, hi) = min(max(e, lo), hi)."""
    assert float(run("fn t() -> f32 { return mix(2.0, 10.0, 0.25); }")) == 4.0
    assert vals(run("fn t() -> vec3f { return clamp(vec3f(-1.0, 0.5, 7.0), vec3f(0.0), vec3f(1.0)); }")) == [0.0, 0.5, 1.0]
    assert vals(run("fn t() -> vec4f { return vec4f(min(1.0, 2.0), max(1.0, 2.0), abs(-3.5), sign(-2.0)); }")) == [1.0, 2.0, 3.5, -1.0]
    assert vals(run("fn t() -> vec3f { return floor(vec3f(-0.5, 0.5, 2.0)); }")) == [-1.0, 0.0, 2.0]


def test_round_is_ties_to_even():
    """round: "ties to even" -- round(2.5) = 2, round(3.5) = 4, round(-0.5) = -0."""
    assert vals(run("fn t() -> vec3f { return round(vec3f(2.5, 3.5, -0.5)); }")) == [2.0, 4.0, -0.0]


# ------------------------------------------------------------------ textures (spec "Texture Built-in Functions"; WebGPU "Sampling")

def _ramp(w, h):
    t = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        for x in range(w):
            t[y, x] = (x, y, 10 * x + y, 1.0)
    return t


SAMPLE = """@group(0) @binding(0) var tex : texture_2d<f32>;
@group(0) @binding(1) var smp : sampler;
fn t(uv : vec2f) -> vec4f { return textureSampleLevel(tex, smp, uv, 0.0); }"""


def test_texture_sample_texel_centres_and_bilinear_weights():
    """WebGPU "Texture sampling": texel (i, j) is centred at ((i + 0.5) / W, (j + 0.5) / H); a linear filter weights the
    four nearest texels by the fractional distance from their centres."""
    res = {"tex": wi.Texture(_ramp(4, 2), "linear", "clamp"), "smp": "sampler"}
    assert vals(run(SAMPLE, args=[wi.Vec([f(1.5 / 4), f(0.5 / 2)])], resources=res)) == [1.0, 0.0, 10.0, 1.0]          # a centre
    assert vals(run(SAMPLE, args=[wi.Vec([f(2.0 / 4), f(0.5 / 2)])], resources=res)) == [1.5, 0.0, 15.0, 1.0]          # between two centres
    assert vals(run(SAMPLE, args=[wi.Vec([f(2.0 / 4), f(1.0 / 2)])], resources=res)) == [1.5, 0.5, 15.5, 1.0]          # between four
    assert vals(run(SAMPLE, args=[wi.Vec([f(1.75 / 4), f(0.5 / 2)])], resources=res)) == [1.25, 0.0, 12.5, 1.0]        # 3/4 : 1/4


def test_texture_address_modes():
    """WebGPU GPUAddressMode: clamp-to-edge repeats the edge texel outside [0, 1] (renderer.ts:77-85, the environment);
    repeat wraps (fullscreen.ts:49-57, the de-noise taps)."""
    clamp = {"tex": wi.Texture(_ramp(4, 2), "linear", "clamp"), "smp": "sampler"}
    rep = {"tex": wi.Texture(_ramp(4, 2), "linear", "repeat"), "smp": "sampler"}
    assert vals(run(SAMPLE, args=[wi.Vec([f(0.0), f(0.25)])], resources=clamp)) == [0.0, 0.0, 0.0, 1.0]
    assert vals(run(SAMPLE, args=[wi.Vec([f(1.0), f(0.25)])], resources=clamp)) == [3.0, 0.0, 30.0, 1.0]
    assert vals(run(SAMPLE, args=[wi.Vec([f(0.0), f(0.25)])], resources=rep)) == [1.5, 0.0, 15.0, 1.0]           # half texel 3, half texel 0
    assert vals(run(SAMPLE, args=[wi.Vec([f(1.0 + 1.5 / 4), f(0.25)])], resources=rep)) == [1.0, 0.0, 10.0, 1.0]  # one period on
    near = {"tex": wi.Texture(_ramp(4, 2), "nearest", "clamp"), "smp": "sampler"}
    assert vals(run(SAMPLE, args=[wi.Vec([f(0.74), f(0.9)])], resources=near)) == [2.0, 1.0, 21.0, 1.0]


def test_texture_load_and_store():
    """Spec "textureLoad" / "textureStore": integer texel coordinates, no filtering.  raytrace.wgsl:477, accumulate.wgsl:20-28."""
    src = """@group(0) @binding(0) var src_tex : texture_2d<f32>;
    @group(0) @binding(1) var dst_tex : texture_storage_2d<rgba16float, write>;
    fn t(p : vec2u) { let c = textureLoad(src_tex, p, 0); textureStore(dst_tex, p, vec4f(c.rgb * 2.0, 1.0)); }"""
    dst = wi.Texture(np.zeros((2, 4, 4), np.float32))
    run(src, args=[wi.Vec([u(3), u(1)])], resources={"src_tex": wi.Texture(_ramp(4, 2)), "dst_tex": dst})
    assert dst.stores == {(3, 1): [6.0, 2.0, 62.0, 1.0]}
