"""PRL front-end (pine_amd/host/prl.cpp, include/pine_prl.h).

CPU part: literal and constant-expression semantics against the REAL reference's psl::stof / stoi /
to_string and vecmath (tests/golden/prl_semantics.txt, written by `oracle/_ref/pine_ref prl`), the
grammar's documented quirks (jit.cpp:1772-1820), overload resolution, control flow, error behaviour,
and that a script builds exactly the scene the API builds from the same values.
GPU part: a script rendered through the front-end equals the oracle's render of the scene it built."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_bit_equal


def _golden():
    lits, exprs = {}, {}
    for line in open(os.path.join(GOLDEN, "prl_semantics.txt")):
        t = line.split()
        if t[0] == "literal":
            lits[t[1]] = (t[2], t[3], t[5])
        elif t[0] == "expr":
            exprs[t[1]] = " ".join(t[2:])
    return lits, exprs


def test_abi_exports_every_declared_symbol():
    from pine_amd import prl
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pine_prl.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(pine_prl_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(prl.SIGNATURES) and len(names) == 5
    for n in names:
        assert hasattr(prl.lib, n)


def test_literals_match_the_reference_psl_conversions():
    from pine_amd import prl
    lits, _ = _golden()
    assert len(lits) >= 30
    for text, (typ, value, as_str) in lits.items():
        assert prl.evaluate(text) == f"{typ} {value}", text
        assert prl.evaluate(f'"" + {text}') == f"str {as_str}", text  # @convert.<T>.str == psl::to_string
    # the point of restating psl::stof: it is NOT strtof
    assert prl.evaluate("0.64") == "f32 0x1.47ae16p-1" and float.fromhex("0x1.47ae16p-1") != np.float32(0.64)


def test_constant_expressions_match_the_reference_vecmath():
    from pine_amd import prl
    _, exprs = _golden()
    assert len(exprs) >= 18
    for text, expect in exprs.items():
        assert prl.evaluate(text) == expect, text


def test_precedence_table_quirks():
    """jit.cpp:1772-1792: codes with a leading 0 are octal, `* / % ^` are decimal; the highest code
    reduces first, first occurrence on ties: `-` before `+`, `/` before `*`, `%`/`^` before `/`."""
    from pine_amd import prl
    ev = prl.evaluate
    assert ev("10 - 4 + 3") == "i32 9"      # (10 - 4) + 3
    assert ev("10 + 4 - 3") == "i32 11"     # 10 + (4 - 3)
    assert ev("7 / 2 * 2") == "i32 6"       # (7 / 2) * 2
    assert ev("2 * 7 / 2") == "i32 6"       # 2 * (7 / 2): `/` outranks `*`
    assert ev("2 * 3 ^ 2") == "i32 18" and ev("2 ^ 3 * 2") == "i32 16"
    assert ev("8 - 2 - 1") == "i32 5"       # equal codes: leftmost first
    assert ev("1 + 2 * 3") == "i32 7" and ev("(1 + 2) * 3") == "i32 9"
    assert ev("1 + 2 < 4 && 2 > 1") == "bool true"
    assert ev("7 % 4 + 1") == "i32 4"
    assert ev("-2 * 3") == "i32 -6" and ev("2 * -3") == "i32 -6"
    assert ev("-0.6") == "f32 -0x1.333334p-1"


def test_typing_and_overload_resolution():
    from pine_amd import prl
    ev = prl.evaluate
    assert ev("[1, 2, 3]") == "vec3i 1 2 3" and ev("[1, 2.0, 3]").startswith("vec3 ")   # jit.cpp:1014-1023
    assert ev("1 + 2.5") == "f32 0x1.cp+1" and ev("7 / 2") == "i32 3" and ev("7 / 2.0") == "f32 0x1.cp+1"
    assert ev("1 < 2.5") == "bool true"          # <(f32, f32) after ONE i32 -> f32 conversion
    assert ev("[1, 2] + [3, 4.0]") == "vec2 0x1p+2 0x1.8p+2"
    assert ev("i32(2.9)") == "i32 2" and ev("f32(3)") == "f32 0x1.8p+1"
    assert ev("[256, 256] * 4") == "vec2i 1024 1024" and ev("vec3(2.0)") == "vec3 0x1p+1 0x1p+1 0x1p+1"
    assert ev("X") == "vec3 0x1p+0 0x0p+0 0x0p+0" and ev("Pi") == "f32 0x1.921fb6p+1"
    assert ev("[1.5, 2.5, 3.5].y") == "f32 0x1.4p+1" and ev("[4, 5, 6].z") == "i32 6"
    assert ev("true + 1") == "str true1"         # +(str, str) with two conversions is the only candidate (context.cpp:143-189)
    for bad, msg in [("1.5 % 2.0", "is not found"), ("Diffuse(0.5)", "is not found"), ("nope(1)", "is not found"),
                     ("[1]", "Only 2, 3, or 4 items"), ("true * 2", "is not found"), ("undefined_name", "is not found"),
                     ("1 +", "primary expression"), ("12345678901234567890", "too large")]:
        with pytest.raises(prl.PrlError, match=msg):
            ev(bad)


def test_statements_and_control_flow():
    from pine_amd import prl
    out = prl.interpret('''
        # declarations, assignment-if-exists, compound assignment
        a := 1; b = 2; b = b + a; a += 4; a *= 2;
        println(a); println(b);
        x := 0.5; x += 1; println(x);
        # ranges: a..b is half-open with ++, a ~ s ~ b is inclusive with += s   (jit.cpp:1519-1546)
        s := 0; for i in 0..5 { s += i; } println(s);
        t := 0.0; for v in 0.0 ~ 0.25 ~ 1.0 { t += v; } println(t);
        n := 0; for k := 10; k > 0; k -= 3 { n++; } println(n);   # C-style: no parentheses (jit.cpp:1548-1556)
        w := 0; while w < 100 { w += 7; if w > 20 { break; } } println(w);
        c := 0; for i in 0..6 { if i % 2 == 0 { continue; } c += i; } println(c);
        if a > 100 { println("big"); } else if a == 10 { println("ten"); } else { println("other"); }
        { shadow := 1; shadow += 1; }
        v := [1, 2, 3] * 2; println(v); println([0.5, 0.25]);
        println(true && false || true); println("a" + "b" + 3);
    ''', dry_run=True)
    assert out.split("\n") == ["10", "3", "1.5000", "10", "2.5000", "4", "21", "9", "ten", "[2 4 6]", "[0.5000 0.2500]",
                               "true", "ab3", ""]
    for src, msg in [("x := 1; y := x +;", "primary expression"), ("break;", "loop"), ("a := 1 a := 2;", "Expect `;`"),
                     ("class A { }", "not supported"), ("s := Scene(); s.render(3);", "is not found"),
                     ("for i in 0..3 { undefined_thing; }", "is not found"), ("PathIntegrator(BlueSampler(4), 0);", "max_path_length"),
                     ("BlueSampler(0);", "positive samples"), ("s := Scene(); s.add(Rect([0,0,0],[1,0,0],[0,0,1]), \"m\");", "Can't find material")]:
        with pytest.raises(prl.PrlError, match=msg):
            prl.interpret(src, dry_run=True)


def _cornell(size, spp, depth):
    src = open(os.path.join(ROOT, "examples", "cornell.pine")).read()
    src = src.replace("size := [640, 640];", f"size := [{size[0]}, {size[1]}];").replace("spp := 16;", f"spp := {spp};")
    return src.replace("depth := 4;", f"depth := {depth};")


def test_script_builds_the_scene_the_api_builds_from_the_same_values():
    """examples/cornell.pine through the front-end (dry run) == the Python API fed with the reference's
    psl::stof values and reference-evaluated constant expressions (independent construction path)."""
    import pine_amd as pa
    from pine_amd import prl
    lits, exprs = _golden()
    F = lambda t: float.fromhex(lits[t][1])  # noqa: E731
    V = lambda key: [float.fromhex(x) for x in exprs[key].split()[1:]]  # noqa: E731
    out = prl.interpret(_cornell((640, 640), 16, 4), dry_run=True)
    ps, spp, depth = prl.scene_of_dry_run(out)
    assert (spp, depth) == (16, 4) and out.rstrip().endswith("@save cornell.png 640x640")

    s = pa.Scene()
    s.add("white", pa.Diffuse([F("0.9")] * 3))
    s.add("blue", pa.Diffuse([F("0.2"), F("0.5"), F("0.9")]))
    s.add("red", pa.Diffuse([F("0.9"), F("0.1"), F("0.05")]))
    s.add("green", pa.Diffuse([F("0.2"), F("0.9"), F("0.05")]))
    s.add(pa.Rect([0, 0, 1], [2, 0, 0], [0, 0, 2], True), "white")
    s.add(pa.Rect([0, 2, 1], [2, 0, 0], [0, 0, 2]), "white")
    s.add(pa.Rect([-1, 1, 1], [0, 0, 2], [0, 2, 0], True), "red")
    s.add(pa.Rect([1, 1, 1], [0, 0, 2], [0, 2, 0]), "green")
    s.add(pa.Rect([0, 1, 2], [2, 0, 0], [0, 2, 0], True), "blue")
    unit = pa.AABB([0, 0, 0], [1, 1, 1])
    m0 = pa.mat4(V("translate([0.0,0.0,0.6])*rotate_y(0.4)*scale([0.6,0.6,0.6])"))
    m1 = pa.mat4(V("translate([-0.6,0.0,1.0])*rotate_y(-0.4)*scale([0.6,1.3,0.6])"))
    s.add(pa.Box(unit, m0), "white")
    s.add(pa.Box(unit, m1), "white")
    s.add(pa.Rect([0.0, F("1.9"), 1], [F("0.1"), 0, 0], [0, 0, F("0.1")]), pa.Emissive(V("600*[1.0,0.64,0.185]")))
    s.set(pa.ThinLenCamera(pa.Film([640, 640], pa.Uncharted2()), [0, 0, 0], [0, 0, 1], F("0.4")))
    assert ps == s.describe()


def test_plane_line_cylinder_triangle_constructors():
    """geometry.cpp:922-929 registrations: Plane(vec3,vec3), Line(vec3,vec3,f32), Cylinder(vec3,vec3,f32),
    Triangle(vec3,vec3,vec3) and their conversion to Shape; examples/shapes.pine == the same scene via the API."""
    import pine_amd as pa
    from pine_amd import prl
    out = prl.interpret(open(os.path.join(ROOT, "examples", "shapes.pine")).read(), dry_run=True)
    ps, spp, depth = prl.scene_of_dry_run(out)
    assert (spp, depth) == (16, 5)
    s = pa.Scene()
    s.add("grey", pa.Diffuse([0.75, 0.75, 0.75]))
    s.add("red", pa.Diffuse([0.875, 0.125, 0.125]))
    s.add("steel", pa.Metal([0.875, 0.75, 0.5], 0.125))
    s.add(pa.Plane([0, 0, 0], [0, 1, 0]), "grey")
    s.add(pa.Cylinder([-0.5, 0.25, 1.25], [-0.5, 1.25, 1.25], 0.25), "red")
    s.add(pa.Line([0.125, 0.125, 0.875], [0.75, 0.875, 1.5], 0.0625), "steel")
    s.add(pa.Triangle([-0.25, 0.0, 1.75], [0.5, 0.0, 1.875], [0.125, 1.0, 1.75]), "red")
    s.add(pa.Rect([0.0, 1.875, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
    s.add(pa.Triangle([-1.0, 1.25, 1.0], [-1.0, 1.5, 1.5], [-1.0, 1.5, 0.75]), pa.Emissive([12.0, 4.0, 2.0]))
    s.add(pa.Line([0.875, 0.5, 1.875], [0.875, 1.5, 1.875], 0.03125), pa.Emissive([2.0, 8.0, 14.0]))
    s.set(pa.ThinLenCamera(pa.Film([96, 96]), [0, 1, -4], [0, 1, 0], 0.25))
    assert ps == s.describe()
    with pytest.raises(prl.PrlError, match="positive thickness"):
        prl.interpret('s := Scene(); s.add(Line([0,0,0],[1,0,0],0.0), Diffuse([1,1,1]));', dry_run=True)


def test_script_functions():
    """`fn name(a: T, ...): R { ... }` and `return` (jit.cpp:1695-1721): registered like built-ins, so overload
    resolution and the one-step conversions apply to calls and to the returned value; a body sees only its
    parameters (in the reference it is a separate JIT'd function; top-level variables are locals of main)."""
    from pine_amd import prl
    src = """
fn sq(x: f32): f32 { return x * x; }
fn fact(n: i32): i32 { if n <= 1 { return 1; } return n * fact(n - 1); }
fn first_big(limit: i32): i32 { for i in 0..100 { if i * i > limit { return i; } } return -1; }
fn bump(v: vec3&): void { v = v + [1.0, 0.0, 0.0]; }
fn sq(v: vec3): f32 { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
p := [1.0, 2.0, 3.0];
bump(p);
println("" + sq(3) + " " + fact(5) + " " + first_big(50) + " " + p + " " + sq(p));
"""
    assert prl.interpret(src, dry_run=True).strip() == "9.0000 120 8 [2.0000 2.0000 3.0000] 17.0000"
    for bad, msg in [("fn f(x: f32): f32 { return y; } f(1.0);", "Variable `y` is not found"),
                     ("y := 2.0; fn f(x: f32): f32 { return x * y; } f(1.0);", "Variable `y` is not found"),
                     ("fn f(x: f32): f32 { } f(1.0);", "ended without returning"),
                     ("fn f(x: nosuch): f32 { return 1.0; }", "Type `nosuch` is not found"),
                     ("return 3;", "only be used inside a function"),
                     ("fn f(x: f32): Shape { return x; } f(1.0);", "where `Shape` is declared"),
                     ("fn r(n: i32): i32 { return r(n + 1); } r(0);", "nested more than 200 deep"),
                     ("f := (p: vec3): f32 { return p[0]; };", "not supported")]:
        with pytest.raises(prl.PrlError, match=msg):
            prl.interpret(bad, dry_run=True)
    # examples/functions.pine: functions returning Shape / Material build the scene the API builds directly
    import pine_amd as pa
    out = prl.interpret(open(os.path.join(ROOT, "examples", "functions.pine")).read(), dry_run=True)
    ps, spp, depth = prl.scene_of_dry_run(out)
    s = pa.Scene()
    s.add("floor", pa.Diffuse([0.75, 0.75, 0.75]))
    s.add(pa.Rect([0, 0, 1], [3, 0, 0], [0, 0, 3], True), "floor")
    s.add(pa.Rect([0, 1, 2.5], [3, 0, 0], [0, 2, 0], True), "floor")
    for i in range(5):
        x, h = -1.0 + 0.5 * i, (0.5 if i % 2 == 0 else 1.0)
        s.add(pa.Box(pa.AABB([x, 0.0, 1.0], [x + 0.25, h, 1.25]), pa.translate([0.0, 0.0, 0.0])),
              pa.Diffuse([0.25 + 0.125 * i, 0.5, 0.75 - 0.125 * i]))
    s.add(pa.Rect([0.0, 1.875, 1], [0.5, 0, 0], [0, 0, 0.5]), pa.Emissive([20.0, 18.0, 15.0]))
    s.set(pa.ThinLenCamera(pa.Film([96, 96]), [0, 1, -4], [0, 1, 0], 0.25))
    assert (spp, depth) == (16, 5) and ps == s.describe()


def test_sobol_sampler_in_scripts():
    """sampler.cpp:182-198: SobolSampler(i32) converts to Sampler like BlueSampler; spp() is the argument as given
    (BlueSampler's is rounded up to a power of two and clamped to 256)."""
    from pine_amd import prl
    assert prl.evaluate("SobolSampler(1000).spp()") == "i32 1000"
    assert prl.evaluate("BlueSampler(1000).spp()") == "i32 256"
    assert prl.evaluate("BlueSampler(5).spp()") == "i32 8"
    with pytest.raises(prl.PrlError, match="thread scheduling"):
        prl.interpret("PathIntegrator(UniformSampler(4), 3);", dry_run=True)
    assert prl.evaluate("HaltonSampler(12).spp()") == "i32 12"  # (as given, sampler.h:44-46)
    with pytest.raises(prl.PrlError, match="positive"):
        prl.interpret("PathIntegrator(HaltonSampler(0), 3);", dry_run=True)
    out_h = prl.interpret(_cornell((64, 64), 8, 4).replace("BlueSampler(spp)", "HaltonSampler(spp)"), dry_run=True)
    assert "@render PathIntegrator HaltonSampler 8 max_path_length 4" in out_h
    src = _cornell((64, 64), 32, 4).replace("BlueSampler(spp)", "SobolSampler(spp)")
    assert "SobolSampler(spp)" in src
    out = prl.interpret(src, dry_run=True)
    assert "@render PathIntegrator SobolSampler 32 max_path_length 4" in out


def test_node_expressions_resolve_like_the_reference():
    """node.cpp:29-116: operators over Nodef / Node3f with the one-step conversions from numbers and vectors."""
    from pine_amd import prl
    ev = prl.evaluate
    assert ev("Position() * 2.0") == "Node3f bin *"              # *(Node3f, Nodef): two conversions, unique
    assert ev("2 * Position()") == "Node3f bin *" and ev("UV()[1]") == "Nodef comp"
    assert ev("[0.9, 0.25, 0.2] * abs(Normal())[1] + [0.05, 0.05, 0.05]") == "Node3f bin +"
    assert ev("lerp(Checkerboard(UV(), 0.95), 0.0, 0.4)") == "Nodef bin +"       # lerp(Nodef, Nodef, Nodef)
    assert ev("lerp(Checkerboard(UV()), [1, 0, 0], [0, 0, 1])") == "Node3f bin +"  # lerp(Nodef, Node3f, Node3f)
    assert ev("lerp(0.25, 2.0, 4.0)") == "f32 0x1.4p+1"                           # psl::lerp
    assert ev("-fract(Position())") == "Node3f un -" and ev("Vec3(UV()[0], 0.5, 1)") == "Node3f tovec3"
    assert ev("Glossy([1, 1, 1], 0.2)") == "Glossy " and ev("Metal(Normal(), 0.1)") == "Metal "
    for bad, msg in [("UV()[3]", "0, 1, or 2"), ("Emissive(Position())", "constant colour"), ("Position() + 1.0", "is not found"),
                     ("Checkerboard(0.5)", "is not found"), ("Metal([1,1,1])", "is not found")]:
        with pytest.raises(prl.PrlError, match=msg):
            ev(bad)


def _classic(n, size, spp):
    src = open(os.path.join(ROOT, "examples", "cone_field.pine")).read()
    return src.replace("n := 100;", f"n := {n};").replace("size := [720, 360];", f"size := [{size[0]}, {size[1]}];").replace(
        "BlueSampler(64)", f"BlueSampler({spp})").replace('scene.camera.film().save("classic.png");', "")


def test_classic_script_with_loops_and_node_graphs_builds_a_valid_scene(oracle):
    """examples/cone_field.pine: PRL for-loops place the cones, the floor is a node graph; the scene description
    the front-end produces is accepted and rendered by the (reference-pinned) oracle."""
    from pine_amd import prl
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(_classic(6, (48, 24), 4), dry_run=True))
    assert ps.count("shape cone ") == 36 and ps.count("\nnode ") == 20 and "material floor uber_n" in ps and (spp, depth) == (4, 6)
    film, st = oracle.render(ps, (48, 24), spp, depth)
    assert np.isfinite(film).all() and film[..., :3].mean() > 0.1
    full = prl.scene_of_dry_run(prl.interpret(_classic(100, (720, 360), 64), dry_run=True))[0]
    assert full.count("shape cone ") == 10000


@pytest.mark.gpu
def test_classic_script_render_equals_oracle(oracle):
    from pine_amd import prl
    src = _classic(10, (90, 45), 16)
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(src, dry_run=True))
    prl.interpret(src)
    ref, _ = oracle.render(ps, (90, 45), spp, depth)
    assert_bit_equal(prl.last_film(), ref, "cone_field.pine through the front-end vs oracle")


def _lights_script(size_note=""):
    return open(os.path.join(ROOT, "examples", "minimal_lights.pine")).read()


def test_lights_and_quick_render(oracle):
    """scene.add(Light), scene.set(Sky), quick_render (program_context.cpp:120-124): list order = add order,
    the environment light last; the oracle (pinned to the reference on light scenes) accepts the description."""
    from pine_amd import prl
    out = prl.interpret(_lights_script(), dry_run=True)
    ps, spp, depth = prl.scene_of_dry_run(out)
    assert (spp, depth) == (4, 4) and "camera thinlens 640 480" in ps and out.rstrip().endswith("@save minimal_lights.png 640x480")
    kinds = [l.split()[1] for l in ps.splitlines() if l.startswith("light ")]
    assert kinds == ["directional", "point", "spot"] and "envlight sky" in ps
    with pytest.raises(prl.PrlError, match="invalid falloff angle"):
        prl.interpret('s := Scene(); s.add(SpotLight([0,1,0], [0,-1,0], [1,1,1], 30.0));', dry_run=True)  # degrees, not radians
    small = ps.replace("camera thinlens 640 480", "camera thinlens 40 30")
    film, _ = oracle.render(small, (40, 30), 4, 4)
    assert film[..., :3].mean() > 0.05


@pytest.mark.gpu
def test_lights_script_render_equals_oracle(oracle):
    from pine_amd import prl
    src = _lights_script().replace('world.quick_render([0, 0.5, -5], [0, 0, 0], "minimal_lights.png");',
                                   'world.set(ThinLenCamera(Film([80, 60]), [0, 0.5, -5], [0, 0, 0], 0.5)); PathIntegrator(BlueSampler(16), 5).render(world);')
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(src, dry_run=True))
    prl.interpret(src)
    ref, _ = oracle.render(ps, (80, 60), spp, depth, order="embree")  # (two-argument PathIntegrator: the reference's default accel)
    assert_bit_equal(prl.last_film(), ref, "lights script through the front-end vs oracle")


def test_runaway_scripts_hit_the_step_budget(monkeypatch):
    from pine_amd import prl
    monkeypatch.setenv("PINE_PRL_MAX_STEPS", "200000")
    with pytest.raises(prl.PrlError, match="step budget"):
        prl.interpret("x := 0; while true { x += 1; }", dry_run=True)


def test_save_extension_rule_and_png_writer(tmp_path):
    """fileio.cpp:55-76: the extension is what follows the FIRST dot; unknown -> warning + '.png'."""
    from pine_amd import prl
    out = prl.interpret('f := Film([4, 2]); f.save("out.v2.png");', dry_run=True)
    assert "Unknown format `v2.png`" in out and "@save out.v2.png.png 4x2" in out
    # real save of an unrendered (black) film: host-only path, no GPU involved
    target = tmp_path / "black.png"
    prl.interpret(f'f := Film([5, 3]); f.save("{target}");')
    data = target.read_bytes() if target.exists() else (tmp_path / "black.png.png").read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    import zlib
    idat = data[data.index(b"IDAT") + 4:data.index(b"IEND") - 8]
    raw = zlib.decompress(idat)
    assert len(raw) == 3 * (5 * 4 + 1) and set(raw[1:21:4]) == {0} and raw[4] == 255  # black, opaque


@pytest.mark.gpu
def test_script_render_equals_oracle_of_the_scene_it_built(oracle):
    from pine_amd import prl
    src = _cornell((72, 56), 16, 5).replace('world.camera.film().save("cornell.png");', "")
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(src, dry_run=True))
    prl.interpret(src)
    film = prl.last_film()
    assert film is not None and film.shape == (56, 72, 4)
    # (the two-argument PathIntegrator is PathIntegrator(EmbreeAccel(), ...) in the reference: EmbreeAccel's order)
    ref, _ = oracle.render(ps, (72, 56), spp, depth, order="embree")
    assert_bit_equal(film, ref, "front-end render vs oracle of the same scene description")


@pytest.mark.gpu
@pytest.mark.parametrize("script,save,sampler", [("shapes.pine", "shapes.png", "blue"), ("functions.pine", "functions.png", "blue"),
                                                 ("functions.pine", "functions.png", "sobol")])
def test_example_scripts_render_equal_the_oracle(oracle, script, save, sampler):
    """examples/shapes.pine (Plane / Line / Cylinder / Triangle) and examples/functions.pine (`fn`) on the GPU,
    the latter also with SobolSampler: the film equals the oracle's render of the scene the script built."""
    from pine_amd import prl
    src = open(os.path.join(ROOT, "examples", script)).read().replace(f'scene.camera.film().save("{save}");', "")
    if sampler == "sobol":
        src = src.replace("BlueSampler(16)", "SobolSampler(16)")
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(src, dry_run=True))
    prl.interpret(src)
    film = prl.last_film()
    assert film is not None and film.shape == (96, 96, 4)
    ref, _ = oracle.render(ps, (96, 96), spp, depth, sampler=sampler, order="embree")  # (two-argument PathIntegrator: the reference's default accel)
    assert_bit_equal(film, ref, script)


def test_accel_and_light_sampler_names_of_the_reference():
    """program_context.cpp:47-52, 76-78: `BVH()`, `Embree()`, `Accel`, `UniformLightSampler()`, `LightSampler` and the
    four-argument PathIntegrator a script on real pine can write; Embree() selects EmbreeAccel's own order, BVH() pine-BVH order; the
    two-argument form is unchanged."""
    from pine_amd import prl
    src = _cornell((64, 64), 8, 4)
    four = src.replace("PathIntegrator(BlueSampler(spp), depth)", "PathIntegrator(Embree(), BlueSampler(spp), UniformLightSampler(), depth)")
    assert four != src
    out = prl.interpret(four, dry_run=True)
    assert "@render PathIntegrator BlueSampler 8 max_path_length 4 accel Embree" in out
    assert prl.scene_of_dry_run(out)[0] == prl.scene_of_dry_run(prl.interpret(src, dry_run=True))[0]  # the same scene
    out = prl.interpret(four.replace("Embree()", "BVH()"), dry_run=True)
    assert "max_path_length 4 accel BVH" in out
    assert "accel" not in prl.interpret(src, dry_run=True).split("@render", 1)[1].splitlines()[0]
    with pytest.raises(prl.PrlError, match="positive"):
        prl.interpret("PathIntegrator(Embree(), BlueSampler(4), UniformLightSampler(), 0);", dry_run=True)
    with pytest.raises(prl.PrlError):  # (no such conversion: a sampler is not a light sampler)
        prl.interpret("PathIntegrator(Embree(), BlueSampler(4), BlueSampler(4), 3);", dry_run=True)


@pytest.mark.gpu
def test_script_with_the_embree_accel_renders_the_embree_order_film(oracle, monkeypatch):
    """A script that names its accel: Embree() -> the film of EmbreeAccel's order (== the real reference's EmbreeAccel film,
    tests/test_gpu_parity.py), BVH() -> pine-BVH order; the two-argument constructor is the
    reference's PathIntegrator(EmbreeAccel(), ...) and renders like Embree(); $PINE_PRL_ACCEL=bvh moves it to pine-BVH order."""
    from pine_amd import prl
    src = _cornell((48, 40), 16, 4).replace('world.camera.film().save("cornell.png");', "")
    ps, spp, depth = prl.scene_of_dry_run(prl.interpret(src, dry_run=True))
    near, _ = oracle.render(ps, (48, 40), spp, depth, order="embree")
    pine, _ = oracle.render(ps, (48, 40), spp, depth)
    assert (near.view(np.uint32) != pine.view(np.uint32)).any()
    four = src.replace("PathIntegrator(BlueSampler(spp), depth)", "PathIntegrator(Embree(), BlueSampler(spp), UniformLightSampler(), depth)")
    prl.interpret(four)
    assert_bit_equal(prl.last_film(), near, "PathIntegrator(Embree(), ...) vs the oracle's embree order")
    prl.interpret(four.replace("Embree()", "BVH()"))
    assert_bit_equal(prl.last_film(), pine, "PathIntegrator(BVH(), ...) vs the oracle")
    monkeypatch.delenv("PINE_PRL_ACCEL", raising=False)
    prl.interpret(src)
    assert_bit_equal(prl.last_film(), near, "PathIntegrator(sampler, n) vs the oracle's embree order (the reference's default accel)")
    monkeypatch.setenv("PINE_PRL_ACCEL", "bvh")
    prl.interpret(src)
    assert_bit_equal(prl.last_film(), pine, "PathIntegrator(sampler, n) with PINE_PRL_ACCEL=bvh")
