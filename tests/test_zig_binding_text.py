"""rayz_amd/zig/renderer_hip.zig is the Zig-side drop-in for `Tracer.render()`.  There is no Zig toolchain in this image, so it
has never been compiled — but its TEXT can be held to the C ABI it binds: every `extern struct` must list the fields of the
C struct of the same name in the same order with the same widths, and every `extern "c" fn` must take the parameters the
prototype in include/rayz_hip.h declares.  One reordered field or a float / double mix-up would otherwise ship silently."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZIG = os.path.join(ROOT, "rayz_amd", "zig", "renderer_hip.zig")
HDR = os.path.join(ROOT, "include", "rayz_hip.h")

C_SCALAR = {"uint32_t": ("u", 4), "uint64_t": ("u", 8), "int": ("i", 4), "double": ("f", 8), "float": ("f", 4), "uint8_t": ("u", 1),
            "size_t": ("u", 8)}
Z_SCALAR = {"u32": ("u", 4), "u64": ("u", 8), "c_int": ("i", 4), "f64": ("f", 8), "f32": ("f", 4), "u8": ("u", 1), "usize": ("u", 8)}


def _strip_c(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _strip_zig(text):
    return "\n".join(re.sub(r"//.*$", "", line) for line in text.split("\n"))


def c_structs():
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", _strip_c(open(HDR).read()), flags=re.S):
        assert m.group(1) == m.group(3)
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(const )?(\w+)(\*?) (\w+)(\[(\d+)\])?$", decl)
            assert mm, decl
            const, ty, ptr, name, _, n = mm.groups()
            if ptr:
                fields.append((name, ("ptr", ty, bool(const)), 1))
            else:
                fields.append((name, C_SCALAR[ty], int(n) if n else 1))
        out[m.group(1)] = fields
    return out


def zig_structs():
    out = {}
    for m in re.finditer(r"pub const (\w+) = extern struct \{(.*?)\n\};", _strip_zig(open(ZIG).read()), flags=re.S):
        fields = []
        for decl in re.split(r",\s*\n", m.group(2)):
            decl = " ".join(decl.split()).rstrip(",")
            if not decl:
                continue
            mm = re.match(r"^(\w+): (.+?)( = .+)?$", decl)
            assert mm, decl
            name, ty = mm.group(1), mm.group(2)
            arr = re.match(r"^\[(\d+)\](\w+)$", ty)
            ptr = re.match(r"^\?\[\*\](const )?(\w+)$", ty)
            if arr:
                fields.append((name, Z_SCALAR[arr.group(2)], int(arr.group(1))))
            elif ptr:
                fields.append((name, ("ptr", ptr.group(2), bool(ptr.group(1))), 1))
            else:
                fields.append((name, Z_SCALAR[ty], 1))
        out[m.group(1)] = fields
    return out


def test_every_extern_struct_matches_the_header_field_for_field():
    c, z = c_structs(), zig_structs()
    assert set(z) == {"RayzTexture", "RayzMaterial", "RayzSphere", "RayzTriangle", "RayzSceneDesc", "RayzCameraDesc", "RayzRenderParams",
                      "RayzRenderStats"}
    for name, zf in z.items():
        assert name in c, f"{name}: no such struct in include/rayz_hip.h"
        assert zf == c[name], f"{name}: zig {zf} != C {c[name]}"
    # and the sizes INTEGRATION.md states follow from those fields (natural alignment, no packing)
    def size(fields):
        off, amax = 0, 1
        for _, ty, n in fields:
            w = 8 if ty[0] == "ptr" else ty[1]
            off = (off + w - 1) // w * w + w * n
            amax = max(amax, w)
        return (off + amax - 1) // amax * amax
    assert {k: size(v) for k, v in z.items()} == {"RayzTexture": 48, "RayzMaterial": 24, "RayzSphere": 64, "RayzTriangle": 80, "RayzSceneDesc": 48,
                                                  "RayzCameraDesc": 152, "RayzRenderParams": 56, "RayzRenderStats": 40}


def c_prototypes():
    out = {}
    text = _strip_c(open(HDR).read())
    for m in re.finditer(r"^(?:const )?(\w+\*?) (rayz_hip_\w+)\(([^)]*)\);", text, flags=re.M | re.S):
        ret, name, params = m.groups()
        plist = []
        params = " ".join(params.split())
        if params != "void":
            for prm in params.split(","):
                mm = re.match(r"^(const )?(\w+(?: long)?)(\*{0,2}) (\w+)$", prm.strip())
                assert mm, (name, prm)
                const, ty, stars, pname = mm.groups()
                plist.append((pname, ty, len(stars), bool(const)))
        out[name] = (ret, plist)
    return out


def zig_prototypes():
    out = {}
    text = _strip_zig(open(ZIG).read())
    text = "\n".join(l for l in text.split("\n") if not l.strip().startswith("///"))
    for m in re.finditer(r'extern "c" fn (\w+)\((.*?)\) ([^;]+);', text, flags=re.S):
        name, params, ret = m.groups()
        plist = []
        for prm in params.split(","):
            prm = " ".join(prm.split())
            if not prm:
                continue
            pname, ty = prm.split(": ", 1)
            plist.append((pname, ty))
        out[name] = (ret.strip(), plist)
    return out


def _zig_type_of(ty, stars, const):
    """The Zig spellings that bind a C parameter `[const] ty *...`."""
    z = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "double": "f64", "float": "f32", "uint8_t": "u8"}.get(ty, ty)
    if stars == 0:
        return {z}
    if stars == 2:
        return {f"*?*{z}"}
    c = "const " if const else ""
    # one object (*T / ?*T) or a buffer ([*]T); optional where the C side documents NULL
    return {f"*{c}{z}", f"?*{c}{z}", f"[*]{c}{z}"}


def test_every_extern_fn_matches_the_header_prototype():
    c, z = c_prototypes(), zig_prototypes()
    must = {"rayz_hip_init", "rayz_hip_last_error", "rayz_hip_render", "rayz_hip_render_f64", "rayz_hip_render_multi", "rayz_hip_render_multi_f64",
            "rayz_hip_multi_create", "rayz_hip_multi_render", "rayz_hip_multi_render_f64", "rayz_hip_multi_destroy"}
    assert must <= set(z), must - set(z)
    for name, (zret, zparams) in z.items():
        assert name in c, f"{name} is not declared in include/rayz_hip.h"
        cret, cparams = c[name]
        assert (zret, cret) in {("c_int", "int"), ("[*:0]const u8", "char*")}, (name, zret, cret)
        norm = lambda n: n[: -len("_or_null")] if n.endswith("_or_null") else n  # noqa: E731
        assert [p[0] for p in zparams] == [norm(p[0]) for p in cparams], (name, zparams, cparams)
        for (pn, zty), (cn, cty, stars, const) in zip(zparams, cparams):
            assert zty in _zig_type_of(cty, stars, const), f"{name}({pn}): zig `{zty}` does not bind C `{'const ' if const else ''}{cty}{'*' * stars}`"
            if cn.endswith("_or_null"):  # the header says NULL is allowed: the Zig side must be able to pass it
                assert zty.startswith("?"), (name, pn, zty)
    # the float / double entries differ exactly in rgb_out
    for f32_fn, f64_fn in (("rayz_hip_render", "rayz_hip_render_f64"), ("rayz_hip_render_multi", "rayz_hip_render_multi_f64"),
                           ("rayz_hip_multi_render", "rayz_hip_multi_render_f64")):
        a, b = dict(z[f32_fn][1]), dict(z[f64_fn][1])
        assert a.pop("rgb_out") == "[*]f32" and b.pop("rgb_out") == "[*]f64" and a == b


def test_options_reach_the_fidelity_mode():
    """HipOptions selects precision and traversal, f64 means the reference's tmin, and the enum values are the header's."""
    text = open(ZIG).read()
    hdr = _strip_c(open(HDR).read())
    assert re.search(r"RAYZ_PRECISION_F32 = 0", hdr) and re.search(r"RAYZ_PRECISION_F64 = 1", hdr)
    assert re.search(r"RAYZ_TRAVERSAL_LINEAR = 0", hdr) and re.search(r"RAYZ_TRAVERSAL_BVH = 1", hdr) and re.search(r"RAYZ_TRAVERSAL_AUTO = 2", hdr)
    assert "pub const Precision = enum(u32) { f32 = 0, f64 = 1 };" in text
    assert "pub const Traversal = enum(u32) { flat_list = 0, bvh = 1, auto = 2 };" in text
    assert re.search(r"precision: Precision = \.f32", text) and re.search(r"traversal: Traversal = \.auto", text)
    assert "if (opt.precision == .f64) @as(f64, 1e-10) else @as(f64, 1e-3)" in text
    assert ".precision = @intFromEnum(opt.precision)" in text and ".traversal = @intFromEnum(opt.traversal)" in text
    for fn in ("rayz_hip_render_f64(&scene", "rayz_hip_multi_render_f64(slot", "rayz_hip_render_multi_f64(opt.devices.ptr"):
        assert fn in text, fn
