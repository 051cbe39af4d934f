"""A subset of the Groan Selection Language (src/select/mod.rs, name.rs, numbers.rs) -- the front end that turns a query
such as "@protein", "resname POPC and name P", "serial 1 to 61" or "Protein or r'^Trans'" into the atom indices of a group
(System::group_create, src/system/groups.rs:36-92).  Host-side string work: it only feeds gr_group_create_from_indices.

Covered: resname, name / atomname, resid / resnum, serial, atomnum, group (explicit or bare group names), the macros
@protein @water @ion @dna @rna @membrane, and / && , or / ||, not / !, parentheses, quoted names, regular expressions
r'...', number lists, ranges (`a to b`, `a-b`) and open ranges (`< n`, `<= n`, `> n`, `>= n`).  Binary operators have equal
precedence and associate to the left, like the reference's parser.  Not covered (SelectError "Unsupported"): chain, element
name / symbol, label, `molecule with` (they need data -- chains, elements, bonds, labels -- outside the scope contract).
Errors carry the reference's variant names (src/errors.rs:452-501)."""
import re

import numpy as np


class SelectError(Exception):
    def __init__(self, variant, detail=""):
        super().__init__("SelectError::%s(%r)" % (variant, detail))
        self.variant, self.detail = variant, detail


# residue / atom name tables of the reference's macros (src/select/mod.rs:590-634; "partially based on GROMACS
# share/top/residuetypes.dat"): data, spelled as queries of this language
MACROS = {
    "@membrane": "(resname r'^[A-Za-z]{2}(PA|PC|PE|PG|PS|PI|GL|DG)$' r'^[A-Za-z]{3}TG' r'.+CL' r'^CER' r'.+SM$' TOG APC CPC IPC LPC OPC PPC TPC UPC VPC XNCE DBG1 DPG1 DPG3 DPGS DXG1 DXG3 PNG1 PNG3 XNG1 XNG3 DFGG DFMG DPGG DPMG DPSG FPGG FPMG FPSG OPGG OPMG OPSG CHOA CHOL CHYO BOG DDM DPC EO5 SDS BOLA BOLB CDL0 CDL1 CDL2 CDL DBG3 ERGO HBHT HDPT HHOP HOPR ACA ACN BCA BCN LCA LCN PCA PCN UCA UCN XCA XCN RAMP REMP OANT POPP1 POPP2 POPP3 DOPP1 DOPP2 DOPP3 POP1 POP2 POP3 DOP1 DOP2 DOP3)",
    "@protein": "(resname ABU ACE AIB ALA ARG ARGN ASN ASN1 ASP ASP1 ASPH ASPP ASH CT3 CYS CYS1 CYS2 CYSH DALA GLN GLU GLUH GLUP GLH GLY HIS HIS1 HISA HISB HISH HISD HISE HISP HSD HSE HSP HYP ILE LEU LSN LYS LYSN LYSH MELEU MET MEVAL NAC NME NHE NH2 PHE PHEH PHEU PHL PRO SER THR TRP TRPH TRPU TYR TYRH TYRU VAL PGLU HID HIE HIP LYP LYN CYN CYM CYX DAB ORN HYP NALA NGLY NSER NTHR NLEU NILE NVAL NASN NGLN NARG NHID NHIE NHIP NHISD NHISE NHISH NTRP NPHE NTYR NGLU NASP NLYS NORN NDAB NLYSN NPRO NHYP NCYS NCYS2 NMET NASPH NGLUH CALA CGLY CSER CTHR CLEU CILE CVAL CASN CGLN CARG CHID CHIE CHIP CHISD CHISE CHISH CTRP CPHE CTYR CGLU CASP CLYS CORN CDAB CLYSN CPRO CHYP CCYS CCYS2 CMET CASPH CGLUH)",
    "@water": "(name W OW HW1 HW2 OH2 H1 H2 and resname SOL WAT HOH OHH TIP T3P T4P T5P T3H W TIP3 TIP4 SPC SPCE)",
    "@ion": "(name NA NA+ CL CL- K K+ SOD CLA CA CA2+ MG ZN CU1 CU LI RB CS F BR I OH Cal CAL IB+ and resname ION NA NA+ CL CL- K K+ SOD CLA CA CA2+ MG ZN CU1 CU LI RB CS F BR I OH Cal CAL IB+)",
    "@dna": "(resname DA DG DC DT DA5 DG5 DC5 DT5 DA3 DG3 DC3 DT3 DAN DGN DCN DTN)",
    "@rna": "(resname A U C G RA RU RC RG RA5 RT5 RU5 RC5 RG5 RA3 RT3 RU3 RC3 RG3 RAN RTN RUN RCN RGN)",
}
_UNSUPPORTED = {"chain", "element", "elname", "elsymbol", "label"}
_MAX = np.iinfo(np.int64).max


def _replace_keywords(s):
    """and / or / not / to -> && / || / ! / - outside quotes (select/mod.rs:653-686)"""
    out, i, quoted = [], 0, False
    while i < len(s):
        c = s[i]
        if c in "'\"":
            quoted = not quoted; out.append(c); i += 1; continue
        if quoted or not c.isalpha():
            out.append(c); i += 1; continue
        j = i
        while j < len(s) and (s[j].isalnum() or s[j] == "_"):
            j += 1
        # a word glued to a quote (r'...') is not a keyword
        word = s[i:j]
        out.append({"and": "&&", "or": "||", "not": "!", "to": "-"}.get(word, word) if not (word == "r" and j < len(s) and s[j] == "'") else word)
        i = j
    return "".join(out)


def _split_args(text):
    """words of a token; 'quoted names' and r'regular expressions' are single words"""
    words, i = [], 0
    while i < len(text):
        c = text[i]
        if c.isspace():
            i += 1; continue
        if c == "r" and text[i + 1:i + 2] == "'":
            j = text.index("'", i + 2)
            words.append(("regex", text[i + 2:j])); i = j + 1; continue
        if c in "'\"":
            j = text.index(c, i + 1)
            words.append(("name", text[i + 1:j])); i = j + 1; continue
        j = i
        while j < len(text) and not text[j].isspace() and text[j] not in "'\"":
            j += 1
        words.append(("name", text[i:j])); i = j
    return words


def _names(words):
    out = []
    for kind, w in words:
        if kind == "regex":
            try:
                out.append(re.compile(w))
            except re.error:
                raise SelectError("InvalidRegex", w)
        else:
            out.append(w)
    return out


def _numbers(words):
    """number lists, `a-b` ranges and open ranges (select/numbers.rs) -> inclusive (first, last) pairs"""
    text = " ".join(w for _, w in words)
    toks = re.findall(r"<=|>=|<|>|-|\d+|\S", text)
    ranges, i = [], 0
    def num(k):
        if k >= len(toks) or not toks[k].isdigit():
            raise SelectError("InvalidNumber")
        return int(toks[k])
    while i < len(toks):
        t = toks[i]
        if t in ("<", "<=", ">", ">="):
            n = num(i + 1)
            if t == "<":
                if n == 0: raise SelectError("InvalidNumber")
                ranges.append((0, n - 1))
            elif t == "<=": ranges.append((0, n))
            elif t == ">": ranges.append((n + 1, _MAX))
            else: ranges.append((n, _MAX))
            i += 2
        elif t.isdigit():
            if i + 1 < len(toks) and toks[i + 1] == "-":
                hi = num(i + 2)
                ranges.append((int(t), hi)); i += 3
            else:
                ranges.append((int(t), int(t))); i += 1
        else:
            raise SelectError("InvalidNumber")
    if not ranges:
        raise SelectError("EmptyArgument")
    return ranges


def _parse_token(token):
    words = _split_args(token)
    if not words:
        raise SelectError("MissingArgument")
    kind0, first = words[0]
    rest = words[1:]
    if kind0 == "name":
        if first in ("resname", "name", "atomname", "resid", "resnum", "serial", "atomnum", "group"):
            if not rest:
                raise SelectError("EmptyArgument")
            if first == "resname": return ("resname", _names(rest))
            if first in ("name", "atomname"): return ("name", _names(rest))
            if first in ("resid", "resnum"): return ("resid", _numbers(rest))
            if first == "serial": return ("serial", _numbers(rest))
            if first == "atomnum": return ("atomnum", _numbers(rest))
            return ("group", _names(rest))
        if first == "atomid":
            raise SelectError("DeprecatedKeyword", "'atomid' is a deprecated Groan Selection Language keyword since `groan_rs` v0.9; use 'atomnum' instead")
        if first in _UNSUPPORTED:
            raise SelectError("Unsupported", first)
    return ("group", _names(words))   # bare words: group names


def _find_parenthesis(s, start):
    depth = 0
    for k in range(start, len(s)):
        if s[k] == "(": depth += 1
        elif s[k] == ")":
            depth -= 1
            if depth == 0:
                return k
    raise SelectError("InvalidParentheses")


def _parse(s, start, end):
    """left-to-right, equal precedence for && and ||, unary ! binds to the next operand (select/mod.rs:394-540)"""
    tree, token, unary, binary = None, [], [], None

    def push(node):
        nonlocal tree, binary
        for _ in unary:
            node = ("not", node)
        unary.clear()
        if tree is None:
            if binary is not None: raise SelectError("MissingArgument")
            tree = node
        else:
            if binary is None: raise SelectError("InvalidTokenParentheses")
            tree = (binary, tree, node)
        binary = None

    def flush():
        text = "".join(token)
        token.clear()
        if text.strip():
            push(_parse_token(text))

    i = start
    while i < end:
        c = s[i]
        if c == "r" and s[i + 1:i + 2] == "'" and (i == start or not (s[i - 1].isalnum() or s[i - 1] == "_")):
            j = s.index("'", i + 2)
            token.append(s[i:j + 1]); i = j + 1
        elif c in "'\"":
            j = s.index(c, i + 1)
            token.append(s[i:j + 1]); i = j + 1
        elif c == "(":
            if "".join(token).strip():
                raise SelectError("InvalidTokenParentheses")
            token.clear()
            j = _find_parenthesis(s, i)
            push(_parse(s, i + 1, j)); i = j + 1
        elif c == ")":
            i += 1
        elif c in "&|":
            if s[i + 1:i + 2] != c:
                raise SelectError("InvalidOperator")
            flush()
            if tree is None or binary is not None:
                raise SelectError("MissingArgument")
            binary = "and" if c == "&" else "or"; i += 2
        elif c == "!":
            if "".join(token).strip():
                raise SelectError("InvalidOperator")
            unary.append("not"); i += 1
        else:
            token.append(c); i += 1
    flush()
    if binary is not None or unary:
        raise SelectError("MissingArgument")
    if tree is None:
        raise SelectError("EmptyArgument")
    return tree


def parse_query(query):
    """query -> selection tree (Select::parse_query, select/mod.rs:46-107)"""
    if not query.strip():
        raise SelectError("EmptyQuery")
    if query.count("(") != query.count(")"):
        raise SelectError("InvalidParentheses", query)
    if query.count("'") % 2 or query.count('"') % 2:
        raise SelectError("InvalidQuotes", query)
    s = query
    if "@" in s:
        for m, expansion in MACROS.items():
            s = s.replace(m, expansion)
    if re.search(r"(molecule\s*with|mol\s*with|molwith)", s):
        raise SelectError("Unsupported", "molecule with")
    s = _replace_keywords(s)
    try:
        return _parse(s, 0, len(s))
    except SelectError as e:
        if e.variant in ("InvalidOperator", "MissingArgument", "EmptyArgument", "InvalidParentheses", "InvalidNumber", "InvalidTokenParentheses"):
            raise SelectError(e.variant, query)
        raise
    except ValueError:
        raise SelectError("InvalidQuotes", query)


def _match(values, uniq_cache, patterns):
    """boolean mask of `values` (list of str) matching any exact name or regular expression"""
    uniq, inverse = uniq_cache
    hit = np.zeros(len(uniq), bool)
    for k, u in enumerate(uniq):
        for p in patterns:
            if (p == u) if isinstance(p, str) else (p.search(u) is not None):
                hit[k] = True; break
    return hit[inverse]


def _in_ranges(arr, ranges):
    keep = np.zeros(arr.shape[0], bool)
    for lo, hi in ranges:
        keep |= (arr >= lo) & (arr <= min(hi, _MAX))
    return keep


def evaluate(tree, structure, groups=None):
    """tree -> boolean mask over the atoms of `structure` (n_atoms, resid, atomid, resname, atomname); `groups` maps group
    names to index arrays (what System::group_create consults for group references)"""
    n = structure.n_atoms
    groups = groups or {}
    cache = {}

    def uniq(attr):
        if attr not in cache:
            vals = np.asarray(getattr(structure, attr))
            cache[attr] = np.unique(vals.astype(str), return_inverse=True)
        return cache[attr]

    def ev(t):
        op = t[0]
        if op == "and": return ev(t[1]) & ev(t[2])
        if op == "or": return ev(t[1]) | ev(t[2])
        if op == "not": return ~ev(t[1])
        if op == "resname": return _match(structure.resname, uniq("resname"), t[1])
        if op == "name": return _match(structure.atomname, uniq("atomname"), t[1])
        if op == "resid": return _in_ranges(np.asarray(structure.resid, np.int64), t[1])
        if op == "serial": return _in_ranges(np.asarray(structure.atomid, np.int64), t[1])      # the file's atom numbers
        if op == "atomnum": return _in_ranges(np.arange(1, n + 1, dtype=np.int64), t[1])        # position in the system, from 1
        if op == "group":
            keep = np.zeros(n, bool)
            for p in t[1]:
                if isinstance(p, str):
                    if p not in groups: raise SelectError("GroupNotFound", p)
                    names = [p]
                else:
                    names = [g for g in groups if p.search(g) is not None]
                for g in names:
                    keep[np.asarray(groups[g], np.int64)] = True
            return keep
        raise SelectError("UnknownError", str(op))

    # a regular expression that matches no group is an error only if NO pattern of its group list matched anything
    def check_regex(t):
        if t[0] in ("and", "or"): check_regex(t[1]); check_regex(t[2])
        elif t[0] == "not": check_regex(t[1])
        elif t[0] == "group":
            matched = [p for p in t[1] if isinstance(p, str) or any(p.search(g) for g in groups)]
            if not matched:
                first = next(p for p in t[1] if not isinstance(p, str))
                raise SelectError("NoRegexMatch", first.pattern)
    check_regex(tree)
    return ev(tree)


def select(structure, query, groups=None):
    """atom indices (sorted, uint64) selected by `query`"""
    return np.nonzero(evaluate(parse_query(query), structure, groups))[0].astype(np.uint64)


def group_create(system, name, query, structure):
    """System::group_create (src/system/groups.rs:36-92): the group `name` = the atoms `query` selects; existing groups of the
    system may be referenced by name.  -> True when an existing group was overwritten (the reference's AlreadyExistsWarning)"""
    groups = {g: np.array(list(system.group_container(g)), np.int64) for g in system.group_names()}
    idx = select(structure, query, groups)
    return system.group_create_from_indices(name, idx)
