"""The C# host sources (host/*.cs) cannot be compiled here (no .NET / Unity in the
image): this test parses host/SphNative.cs and checks it against include/sphfluid.h -- every
[StructLayout] struct field for field (name, element type, array length, offset, total size) and
the [DllImport] list against the header's SPH_API list -- and that FluidSPHByHIPSystem.cs only
calls entry points that the binding declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_TYPES = {"float": ("f32", 4), "int32_t": ("i32", 4), "double": ("f64", 8), "int64_t": ("i64", 8)}
CS_TYPES = {"float": ("f32", 4), "int": ("i32", 4), "double": ("f64", 8), "long": ("i64", 8)}


def _layout(fields):
    """fields: (name, kind, size, count) -> [(name, kind, count, offset)], total size"""
    off, align, out = 0, 1, []
    for name, kind, size, count in fields:
        off = (off + size - 1) // size * size
        out.append((name, kind, count, off))
        off += size * count
        align = max(align, size)
    return out, (off + align - 1) // align * align


def c_structs():
    text = open(os.path.join(ROOT, "include", "sphfluid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    consts = {m.group(1): None for m in re.finditer(r"\b(SPH_T_COUNT)\b", text)}
    enum = re.search(r"enum\s*\{\s*SPH_T_KEYS\s*=\s*0(.*?)\}", text, re.S).group(1)
    consts["SPH_T_COUNT"] = len([x for x in enum.split(",") if x.strip()])  # KEYS + the rest - COUNT
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.fullmatch(r"(\w+)\s+(\w+)(?:\[(\w+)\])?", decl)
            assert fm, decl
            kind, size = C_TYPES[fm.group(1)]
            n = fm.group(3)
            count = 1 if n is None else (int(n) if n.isdigit() else consts[n])
            fields.append((fm.group(2), kind, size, count))
        out[m.group(1)] = _layout(fields)
    return out


def cs_structs():
    text = open(os.path.join(ROOT, "host", "SphNative.cs")).read()
    text = re.sub(r"//.*", "", text)
    out = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public struct (\w+)\s*\{(.*?)\}",
                         text, re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.fullmatch(r"public\s+(fixed\s+)?(\w+)\s+(\w+)(?:\[(\d+)\])?", decl)
            assert fm, decl
            kind, size = CS_TYPES[fm.group(2)]
            assert (fm.group(1) is None) == (fm.group(4) is None), decl
            fields.append((fm.group(3), kind, size, int(fm.group(4) or 1)))
        out[m.group(1)] = _layout(fields)
    return out


def test_csharp_structs_match_the_header():
    c, cs = c_structs(), cs_structs()
    assert sorted(c) == sorted(cs), (sorted(c), sorted(cs))
    for name in c:
        assert c[name] == cs[name], name
    assert c["SphBox"][1] == 100 and c["SphParticle"][1] == 16 and c["SphBoundaryParticle"][1] == 12


def test_csharp_timing_count_matches():
    text = open(os.path.join(ROOT, "host", "SphNative.cs")).read()
    n = int(re.search(r"SPH_T_COUNT = (\d+)", text).group(1))
    assert c_structs()["SphTimings"][0][0][2] == n


def _header_exports():
    text = open(os.path.join(ROOT, "include", "sphfluid.h")).read()
    return sorted(set(re.findall(r"SPH_API\s+[\w\s\*]+?\b(sph_\w+)\s*\(", text)))


def _header_argc():
    text = open(os.path.join(ROOT, "include", "sphfluid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"SPH_API\s+[\w\s\*]+?\b(sph_\w+)\s*\((.*?)\)\s*;", text, re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_csharp_dllimports_cover_the_header():
    text = open(os.path.join(ROOT, "host", "SphNative.cs")).read()
    decls = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern \w+\*? (sph_\w+)\((.*?)\);", text)
    assert sorted(d[0] for d in decls) == _header_exports()
    argc = _header_argc()
    for name, args in decls:
        assert len(args.split(",")) == argc[name], name


def test_host_system_calls_only_declared_entry_points():
    text = open(os.path.join(ROOT, "host", "FluidSPHByHIPSystem.cs")).read()
    used = set(re.findall(r"SphNative\.(sph_\w+)\(", text))
    assert used and used <= set(_header_exports()), used - set(_header_exports())
    assert "sph_step_compat" in used and "sph_step" in used and "sph_set_colliders" in used
