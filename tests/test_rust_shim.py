"""rust/src/groan_hip.rs against include/groan_hip.h, mechanically.

There is no Rust toolchain in the build image, so the groan_rs-side binding has never been compiled.  What CAN be checked here is what
`bindgen` would have guaranteed: every `extern "C"` declaration of the shim names a function the header declares, with the same number
of parameters, and every parameter and the return value agree in width, pointer depth and the constness of what is pointed at; the
`pub const GR_*` values equal the header's enumerators; `#[repr(C)]` structs have the header's fields in the header's order.  The
pattern is the crate's own xdrfile FFI (src/io/xdrfile.rs:27-120).  This is not `cargo check`; it is what can be done without one."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return re.sub(r"//[^\n]*", " ", txt)


def _strip_rs(txt):
    txt = re.sub(r"//[^\n]*", " ", txt)                 # line comments first: they may mention paths like src/io/xtc_io/*
    return re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)


C_SCALARS = {"int": ("i", 32), "unsigned": ("u", 32), "uint32_t": ("u", 32), "int32_t": ("i", 32), "uint64_t": ("u", 64), "int64_t": ("i", 64),
             "size_t": ("u", "size"), "float": ("f", 32), "double": ("f", 64), "char": ("c", 8), "void": ("v", 0), "uint8_t": ("u", 8)}
RS_SCALARS = {"c_int": ("i", 32), "i32": ("i", 32), "u32": ("u", 32), "u64": ("u", 64), "i64": ("i", 64), "usize": ("u", "size"), "c_float": ("f", 32),
              "f32": ("f", 32), "f64": ("f", 64), "c_double": ("f", 64), "c_char": ("c", 8), "c_void": ("v", 0), "u8": ("u", 8)}


def _c_type(t, fn_types):
    """canonical form of a C parameter / return type: (kind, width, [constness of each pointer level's pointee, outermost first])"""
    t = t.strip()
    t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*(\[[^\]]*\])+\s*$", lambda m: "*" * m.group(0).count("["), t) if re.search(r"\[[^\]]*\]\s*$", t) else t
    if re.search(r"\(\s*\*", t):                        # inline function pointer
        return ("fn", 0, [])
    depth = t.count("*")
    head = t.split("*")[0]
    const_pointee = bool(re.search(r"\bconst\b", head))
    words = [w for w in re.split(r"\s+", re.sub(r"\bconst\b|\bstruct\b", " ", t.replace("*", " "))) if w]
    # the last word is the parameter name when there are two or more words and the last is not a type
    known = set(C_SCALARS) | fn_types
    name_words = words[:-1] if len(words) >= 2 and (words[-1] not in known or words[-2] in known or words[-2].startswith("gr_")) else words
    if len(words) >= 2 and words[-1] in known and words[-2] in ("unsigned",):
        name_words = words
    base = " ".join(name_words)
    base = {"unsigned int": "unsigned", "unsigned long long": "uint64_t", "long long": "int64_t"}.get(base, base)
    if base in fn_types:
        return ("fn", 0, [])
    if base in C_SCALARS:
        kind, width = C_SCALARS[base]
    else:
        kind, width = ("s:" + base, 0)                  # a struct / opaque handle by name
    return (kind, width, [const_pointee] + [False] * (depth - 1) if depth else [])


def _rs_type(t, fn_types):
    t = t.strip()
    consts = []
    while True:
        m = re.match(r"\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        consts.append(m.group(1) == "const")
        t = m.group(2).strip()
    if t in fn_types or t.startswith("Option<"):
        return ("fn", 0, [])
    if t in RS_SCALARS:
        kind, width = RS_SCALARS[t]
    else:
        kind, width = ("s:" + t, 0)
    # Rust spells the constness of every level; C code here only marks the innermost pointee.  Compare the level that matters: data const or not.
    return (kind, width, [consts[-1]] + [False] * (len(consts) - 1) if consts else [])


def _header():
    txt = _strip_c(open(os.path.join(ROOT, "include", "groan_hip.h")).read())
    fn_types = set(re.findall(r"typedef\s+[^;(]*\(\s*\*\s*([A-Za-z_0-9]+)\s*\)\s*\([^;]*\)\s*;", txt))
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)\b(gr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        # split parameters at top-level commas (function-pointer parameters carry commas in parentheses)
        parts, depth, cur = [], 0, ""
        for ch in params:
            if ch == "(":
                depth += 1
            if ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur); cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        if len(parts) == 1 and parts[0].strip() == "void":
            parts = []
        protos[name] = (_c_type(ret + " r_", fn_types) if ret.strip() != "void" else ("v", 0, []), [_c_type(p, fn_types) for p in parts])
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(GR_[A-Z0-9_]+)\s*=\s*(-?\d+)", txt)}
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+[A-Za-z_0-9]*\s*\{([^}]*)\}\s*([A-Za-z_0-9]+)\s*;", txt, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty = decl.split()[0]
            for f in decl[len(ty):].split(","):
                fm = re.match(r"\s*([A-Za-z_0-9]+)\s*(\[\s*(\d+)\s*\])?", f)
                fields.append((fm.group(1), C_SCALARS[ty], int(fm.group(3)) if fm.group(3) else 0))
        structs[m.group(2)] = fields
    return protos, enums, structs, fn_types


def _shim():
    txt = _strip_rs(open(os.path.join(ROOT, "rust", "src", "groan_hip.rs")).read())
    fn_types = set(re.findall(r"pub\s+type\s+([A-Za-z_0-9]+)\s*=\s*Option<", txt))
    decls = {}
    for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', txt, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(gr_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
            name, params, ret = m.group(1), m.group(2), m.group(3)
            parts = [p.split(":", 1)[1] for p in re.split(r",(?![^<]*>)", params) if ":" in p]
            decls[name] = (_rs_type(ret, fn_types) if ret else ("v", 0, []), [_rs_type(p, fn_types) for p in parts])
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub\s+const\s+(GR_[A-Z0-9_]+)\s*:\s*c_int\s*=\s*(-?\d+)\s*;", txt)}
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub\s+struct\s+([A-Za-z_0-9]+)\s*\{(.*?)\}", txt, flags=re.S):
        fields = []
        for f in re.split(r",(?![^\[]*\])", m.group(2)):
            fm = re.match(r"\s*(?:pub\s+)?([A-Za-z_0-9]+)\s*:\s*(\[\s*([A-Za-z_0-9]+)\s*;\s*(\d+)\s*\]|[A-Za-z_0-9]+)", f)
            if fm:
                ty = fm.group(3) or fm.group(2)
                if ty in RS_SCALARS:
                    fields.append((fm.group(1), RS_SCALARS[ty], int(fm.group(4)) if fm.group(4) else 0))
        structs[m.group(1)] = fields
    return decls, consts, structs


def test_every_extern_declaration_of_the_shim_matches_the_header():
    protos, _, _, _ = _header()
    decls, _, _ = _shim()
    assert len(protos) >= 100 and len(decls) >= 80, (len(protos), len(decls))
    bad = []
    for name, (ret, params) in sorted(decls.items()):
        if name not in protos:
            bad.append("%s: not declared in include/groan_hip.h" % name)
            continue
        cret, cparams = protos[name]
        if len(params) != len(cparams):
            bad.append("%s: %d parameters in the shim, %d in the header" % (name, len(params), len(cparams)))
            continue
        for k, (a, b) in enumerate([(ret, cret)] + list(zip(params, cparams))):
            what = "return value" if k == 0 else "parameter %d" % k
            if a[0] != b[0] or a[1] != b[1]:
                bad.append("%s, %s: shim %r, header %r" % (name, what, a[:2], b[:2]))
            elif len(a[2]) != len(b[2]):
                bad.append("%s, %s: pointer depth %d in the shim, %d in the header" % (name, what, len(a[2]), len(b[2])))
            elif a[2] and a[2][0] != b[2][0]:
                bad.append("%s, %s: pointee is %s in the shim, %s in the header" % (name, what, "const" if a[2][0] else "mut", "const" if b[2][0] else "mutable"))
    assert not bad, "\n".join(bad)


def test_constants_and_repr_c_structs_of_the_shim_are_the_header_s():
    _, enums, cstructs, _ = _header()
    _, consts, rstructs = _shim()
    assert len(consts) >= 8
    for name, v in consts.items():
        assert name in enums and enums[name] == v, (name, v, enums.get(name))
    # every status the shim's error mapping matches on exists (it names them in `match` arms)
    txt = open(os.path.join(ROOT, "rust", "src", "groan_hip.rs")).read()
    for name in set(re.findall(r"\bGR_(?:E|OK|DIM|CENTER|PROGRESS)_?[A-Z0-9_]*\b", txt)):
        assert name in enums or name in consts, name
    for name, fields in rstructs.items():
        if not fields or [f for f, _, _ in fields] == ["_private"]:     # opaque handles: `_private: [u8; 0]`
            continue
        assert name in cstructs, name
        assert [(f, t, n) for f, t, n in fields] == [(f, t, n) for f, t, n in cstructs[name]], (name, fields, cstructs[name])


def test_the_shim_binds_the_whole_hot_path():
    """the calls INTEGRATION.md routes groan_rs through must be bound: a header function the shim forgets is a call site that cannot be written"""
    protos, _, _, _ = _header()
    decls, _, _ = _shim()
    need = ["gr_ctx_create", "gr_ctx_destroy", "gr_set_masses", "gr_frame_upload", "gr_frame_download", "gr_group_create_from_ranges", "gr_group_center",
            "gr_group_distance", "gr_group_all_distances", "gr_group_translate", "gr_group_wrap", "gr_atoms_center", "gr_rmsd_plan_create", "gr_rmsd_plan_destroy",
            "gr_rmsd_batch", "gr_rmsd_fit_batch", "gr_calc_rmsd", "gr_calc_rmsd_and_fit", "gr_pool_create", "gr_pool_map_range", "gr_comm_create",
            "gr_comm_gather_per_frame", "gr_xtc_open", "gr_xtc_read_frames_device", "gr_ctx_set_tuning", "gr_ctx_stat"]
    for n in need:
        assert n in protos and n in decls, n
