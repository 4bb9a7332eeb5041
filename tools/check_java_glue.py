#!/usr/bin/env python3
"""Static check of the reference-side Java glue (java/src) against the reference's sources -- the build image has no JDK,
so nothing compiles these files; this script is the next best thing and runs in the test-suite (build container only:
it reads /root/reference).

For every class under java/src it checks, against an index of the reference's declarations
(core|base/src/main/java):
  * `extends X` / `implements X`: X exists, is visible, is not final;
  * every @Override method: an ancestor declares a method of that name and arity that is neither private, final nor
    static, and is visible from the glue class's package;
  * every `super(...)` call: an ancestor constructor of that arity exists and is visible;
  * every unqualified identifier that is not declared in the glue class itself but names a field or method of an
    ancestor: the member is visible from the subclass (private and foreign package-private members are errors);
  * every `x.member` / `x.member(...)` where x has a declared reference type (local, parameter, field, inherited field,
    or the result of a previous member in a chain): the member exists on that type (or its ancestors / interfaces) and
    is visible; `new X(...)` of a reference class: a constructor of that arity exists; `X.CONSTANT` on reference
    classes and enums.
Types that come from the JDK are skipped.  It is a name / arity / visibility check, not a type checker.

    python tools/check_java_glue.py [--reference /root/reference] [--glue java/src]      exit code 1 on findings
"""
import argparse
import os
import re
import sys

KEYWORDS = {"if", "else", "for", "while", "do", "switch", "case", "return", "new", "try", "catch", "finally", "throw",
            "throws", "this", "super", "null", "true", "false", "instanceof", "int", "long", "double", "float", "boolean",
            "byte", "short", "char", "void", "final", "static", "public", "protected", "private", "abstract", "class",
            "interface", "enum", "extends", "implements", "import", "package", "synchronized", "volatile", "transient",
            "native", "break", "continue", "default", "assert", "var"}
MODS = {"public", "protected", "private", "static", "final", "abstract", "synchronized", "volatile", "transient", "native",
        "default", "strictfp"}


def strip_comments(src):
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            seg = src[i:(n if j < 0 else j + 2)]
            out.append("\n" * seg.count("\n"))
            i = n if j < 0 else j + 2
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("'x'")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def strip_generics(s):
    out, depth = [], 0
    for c in s:
        if c == "<":
            depth += 1
        elif c == ">":
            depth = max(0, depth - 1)
        elif depth == 0:
            out.append(c)
    return "".join(out)


def split_top(s, sep=","):
    parts, depth, cur = [], 0, []
    for c in s:
        if c in "(<[{":
            depth += 1
        elif c in ")>]}":
            depth -= 1
        if c == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(c)
    if "".join(cur).strip():
        parts.append("".join(cur))
    return parts


class Cls:
    def __init__(self, name, package, outer, mods, kind, extends, implements, imports, path):
        self.name, self.package, self.outer, self.mods, self.kind = name, package, outer, mods, kind
        self.extends, self.implements, self.imports, self.path = extends, implements, imports, path
        self.fields = {}    # name -> (mods, type)
        self.methods = {}   # name -> [(mods, [param types], return type)]
        self.ctors = []     # [(mods, [param types])]
        self.bodies = []    # glue only: (kind, name, params [(type, name)], body text, has_override, line)

    @property
    def qual(self):
        return (self.outer.qual + "." if self.outer else (self.package + "." if self.package else "")) + self.name


def visibility(mods):
    for v in ("public", "protected", "private"):
        if v in mods:
            return v
    return "package"


def parse_params(s):
    res = []
    for p in split_top(s):
        p = strip_generics(p).replace("final ", " ").strip()
        p = re.sub(r"@\w+(\([^)]*\))?\s*", "", p)
        if not p:
            continue
        toks = p.split()
        res.append((" ".join(toks[:-1]).replace(" ", ""), toks[-1]))
    return res


def parse_file(path, keep_bodies=False):
    return parse_source(strip_comments(open(path, errors="replace").read()), path, keep_bodies)


def parse_source(src, path, keep_bodies=False):
    package = (re.search(r"\bpackage\s+([\w.]+)\s*;", src) or [None, ""])[1]
    imports = re.findall(r"\bimport\s+(?:static\s+)?([\w.]+(?:\.\*)?)\s*;", src)
    classes, stack = [], []   # stack of (ctx kind, Cls or None)
    i, n, start = 0, len(src), 0
    line_of = lambda pos: src.count("\n", 0, pos) + 1
    pending_override = False
    while i < n:
        c = src[i]
        if c == "{":
            raw_header = src[start:i].strip()
            header = re.sub(r"@\w+(\s*\([^()]*\))?", " ", raw_header).strip()   # annotations carry no declaration
            m = re.search(r"((?:[\w@]+(?:\([^)]*\))?\s+)*)(class|interface|enum)\s+(\w+)\s*(<[^{]*?>)?\s*(extends\s+[^{]*?)?(implements\s+[^{]*)?$", header, re.S)
            top = stack[-1] if stack else None
            if m and (top is None or top[0] == "class"):
                mods = set(re.findall(r"\b(\w+)\b", m.group(1))) & MODS
                ext = strip_generics(m.group(5) or "").replace("extends", "").strip()
                imp = [x.strip() for x in split_top(strip_generics(m.group(6) or "").replace("implements", ""))]
                if m.group(2) == "interface":
                    imp, ext = [x.strip() for x in split_top(ext)] if ext else [], ""
                cls = Cls(m.group(3), package, top[1] if top else None, mods, m.group(2), ext, [x for x in imp if x], imports, path)
                if top and top[1].kind == "interface":
                    cls.mods |= {"public", "static"}
                classes.append(cls)
                stack.append(("class", cls))
                if m.group(2) == "enum":  # constants up to the first ';' or the closing brace
                    j, depth = i + 1, 0
                    while j < n and not (depth == 0 and src[j] in ";}"):
                        depth += src[j] in "({"
                        depth -= src[j] in ")}"
                        j += 1
                    for const in split_top(src[i + 1:j]):
                        cm = re.match(r"\s*(?:@\w+(?:\([^)]*\))?\s*)*(\w+)", const)
                        if cm:
                            cls.fields[cm.group(1)] = ({"public", "static", "final"}, cls.name)
                    if j < n and src[j] == ";":
                        i = j
                        start = j + 1
                    else:
                        start = i + 1
                    i += 1
                    continue
            elif top and top[0] == "class" and "(" in header and not re.search(r"=\s*new\b|[^=!<>]=[^=]", strip_generics(header.split("(")[0])):
                cls = top[1]
                hm = re.search(r"((?:[\w@.<>\[\],?\s]|\([^)]*\))*?)\b(\w+)\s*\(([^{]*)\)\s*(?:throws\s+[\w.,\s]+)?$", header, re.S)
                if hm:
                    pre = strip_generics(re.sub(r"@\w+(\([^)]*\))?", " ", hm.group(1)))
                    toks = pre.split()
                    mods = set(toks) & MODS
                    rtype = " ".join(t for t in toks if t not in MODS)
                    if cls.kind == "interface" and "private" not in mods:
                        mods.add("public")
                    params = parse_params(hm.group(3))
                    name = hm.group(2)
                    override = "@Override" in raw_header
                    if name == cls.name and not rtype:
                        cls.ctors.append((mods, [p[0] for p in params]))
                        kind = "ctor"
                    else:
                        cls.methods.setdefault(name, []).append((mods, [p[0] for p in params], rtype))
                        kind = "method"
                    if keep_bodies:
                        j, depth = i, 0
                        while j < n:
                            depth += src[j] == "{"
                            depth -= src[j] == "}"
                            if depth == 0:
                                break
                            j += 1
                        cls.bodies.append((kind, name, params, src[i:j + 1], override, line_of(i)))
                stack.append(("body", None))
            else:
                stack.append(("block", None))
            start = i + 1
        elif c == "}":
            if stack:
                stack.pop()
            start = i + 1
        elif c == ";":
            top = stack[-1] if stack else None
            if top and top[0] == "class":
                decl = re.sub(r"@\w+(\s*\([^()]*\))?", " ", src[start:i]).strip()
                cls = top[1]
                if decl and not decl.startswith(("import", "package")):
                    # (a field whose initialiser ends in a call -- `final ByteBuffer b = alloc(n).order(x);` -- is not a method declaration)
                    ends_in_paren = bool(re.search(r"\)\s*(throws[\w.,\s]+)?$", decl)) and "=" not in strip_generics(decl.split("(")[0])
                    head = decl.split("=")[0] if not ends_in_paren else decl
                    if "(" in head and ends_in_paren:   # abstract / interface method
                        hm = re.search(r"((?:[\w@.<>\[\],?\s]|\([^)]*\))*?)\b(\w+)\s*\((.*)\)\s*(?:throws\s+[\w.,\s]+)?$", decl, re.S)
                        if hm:
                            toks = strip_generics(re.sub(r"@\w+(\([^)]*\))?", " ", hm.group(1))).split()
                            mods = set(toks) & MODS
                            if cls.kind == "interface":
                                mods.add("public")
                            cls.methods.setdefault(hm.group(2), []).append(
                                (mods, [p[0] for p in parse_params(hm.group(3))], " ".join(t for t in toks if t not in MODS)))
                    else:
                        first = split_top(strip_generics(re.sub(r"@\w+(\([^)]*\))?", " ", decl)))
                        toks = first[0].split("=")[0].split()
                        mods = set(toks) & MODS
                        rest = [t for t in toks if t not in MODS]
                        if len(rest) >= 2:
                            ftype = "".join(rest[:-1])
                            if cls.kind == "interface":
                                mods |= {"public", "static", "final"}
                            cls.fields[rest[-1].rstrip("[]")] = (mods, ftype)
                            for more in first[1:]:
                                nm = more.split("=")[0].strip()
                                if re.fullmatch(r"\w+(\[\])*", nm):
                                    cls.fields[nm.rstrip("[]")] = (mods, ftype)
            start = i + 1
        i += 1
    return classes


class Index:
    def __init__(self):
        self.by_qual, self.by_simple = {}, {}

    def add(self, classes):
        for c in classes:
            self.by_qual[c.qual] = c
            self.by_simple.setdefault(c.name, []).append(c)

    def resolve(self, name, ctx):
        """class for a (possibly dotted, possibly generic) type name as seen from class ctx, or None (JDK / unknown)"""
        name = strip_generics(name).replace("[]", "").replace("...", "").strip()
        if not name or name in KEYWORDS:
            return None
        if name in self.by_qual:
            return self.by_qual[name]
        parts = name.split(".")
        c = ctx
        while c is not None:   # nested class of ctx, its outers or their ancestors
            for cand in self.ancestors(c, include_self=True):
                q = cand.qual + "." + name
                if q in self.by_qual:
                    return self.by_qual[q]
            c = c.outer
        for imp in ctx.imports:
            if imp.endswith("." + parts[0]):
                q = imp + ("." + ".".join(parts[1:]) if len(parts) > 1 else "")
                if q in self.by_qual:
                    return self.by_qual[q]
            if imp.endswith(".*") and imp[:-2] + "." + name in self.by_qual:
                return self.by_qual[imp[:-2] + "." + name]
        q = (ctx.package + "." if ctx.package else "") + name
        if q in self.by_qual:
            return self.by_qual[q]
        if len(parts) > 1:   # Outer.Inner by simple outer name
            outer = self.resolve(parts[0], ctx)
            if outer is not None and outer.qual + "." + ".".join(parts[1:]) in self.by_qual:
                return self.by_qual[outer.qual + "." + ".".join(parts[1:])]
        return None

    def ancestors(self, cls, include_self=False):
        seen, todo, out = set(), [cls], []
        while todo:
            c = todo.pop(0)
            if c.qual in seen:
                continue
            seen.add(c.qual)
            if c is not cls or include_self:
                out.append(c)
            for t in ([c.extends] if c.extends else []) + c.implements:
                r = self.resolve_raw(t, c)
                if r is not None:
                    todo.append(r)
        return out

    def resolve_raw(self, name, ctx):   # (ancestors() must not recurse through resolve()'s ancestor walk)
        name = strip_generics(name).strip()
        if name in self.by_qual:
            return self.by_qual[name]
        c = ctx
        while c is not None:
            if c.qual + "." + name in self.by_qual:
                return self.by_qual[c.qual + "." + name]
            c = c.outer
        for imp in ctx.imports:
            if imp.endswith("." + name.split(".")[0]):
                q = imp + name[len(name.split(".")[0]):]
                if q in self.by_qual:
                    return self.by_qual[q]
        q = (ctx.package + "." if ctx.package else "") + name
        if q in self.by_qual:
            return self.by_qual[q]
        cands = self.by_simple.get(name.split(".")[-1], [])
        return cands[0] if len(cands) == 1 and "." not in name else None


def accessible(mods, owner, user, via_inheritance):
    v = visibility(mods)
    if v == "public":
        return True
    if v == "private":
        return owner.qual == user.qual or (owner.outer is not None and owner.outer.qual == user.qual)
    if owner.package == user.package:
        return True
    return v == "protected" and via_inheritance


def find_member(idx, cls, name, want):
    """(owner, mods, type, arities) of field/method `name` on cls or its ancestors; want = 'field' | 'method' | 'any'"""
    for c in [cls] + idx.ancestors(cls):
        if want in ("field", "any") and name in c.fields:
            return c, c.fields[name][0], c.fields[name][1], None
        if want in ("method", "any") and name in c.methods:
            ms = c.methods[name]
            return c, ms[0][0], ms[0][2], ms
    return None


def count_args(body, pos):
    """number of top-level arguments of the call whose '(' is at body[pos]"""
    depth, j, commas, any_tok = 0, pos, 0, False
    while j < len(body):
        ch = body[j]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
        elif ch == "," and depth == 1:
            commas += 1
        elif depth >= 1 and not ch.isspace():
            any_tok = True
        j += 1
    return (commas + 1) if any_tok else 0


def arity_ok(overloads, n):
    for ov in overloads:
        params = ov[1]
        if len(params) == n or (params and params[-1].endswith("...") and n >= len(params) - 1):
            return True
    return False


def split_anonymous(body):
    """(body with the anonymous class bodies blanked out, [(supertype, class body text, offset)])"""
    found, out = [], body
    for m in re.finditer(r"\bnew\s+([\w.]+)\s*(?:<[^<>(){};]*>)?\s*\(", body):
        depth, j = 0, m.end() - 1
        while j < len(body):
            depth += body[j] == "("
            depth -= body[j] == ")"
            if depth == 0:
                break
            j += 1
        k = j + 1
        while k < len(body) and body[k].isspace():
            k += 1
        if k < len(body) and body[k] == "{":
            depth, e = 0, k
            while e < len(body):
                depth += body[e] == "{"
                depth -= body[e] == "}"
                if depth == 0:
                    break
                e += 1
            found.append((m.group(1), body[k:e + 1], k))
    for _, text, off in found:
        out = out[:off] + "{" + re.sub(r"[^\n]", " ", text[1:-1]) + "}" + out[off + len(text):]
    return out, found


def check_class(idx, cls, problems, line_shift=0):
    def bad(line, msg):
        problems.append(f"{os.path.relpath(cls.path)}:{line + line_shift}: {cls.name}: {msg}")

    # ---- supertypes
    for t in ([cls.extends] if cls.extends else []) + cls.implements:
        sup = idx.resolve(t, cls)
        if sup is None:
            if t.split(".")[0] not in ("Runnable", "Closeable", "AutoCloseable", "Serializable", "Comparable", "Object"):
                bad(1, f"supertype {t} not found in the reference")
            continue
        if "final" in sup.mods:
            bad(1, f"supertype {sup.qual} is final")
        if not accessible(sup.mods if sup.outer else (sup.mods | ({"public"} if "public" in sup.mods else set())), sup, cls, True) and sup.package != cls.package:
            bad(1, f"supertype {sup.qual} is not visible")
    ancestors = idx.ancestors(cls)
    own_fields = set(cls.fields)
    own_methods = set(cls.methods)
    for kind, name, params, body, override, line in cls.bodies:
        # ---- anonymous subclasses inside the body: checked as classes of their own, nested in this one
        body, anon = split_anonymous(body)
        for n_anon, (sup_name, text, off) in enumerate(anon):
            src = "package %s;\n%s\nclass %s_anon%d extends %s %s" % (cls.package, "\n".join("import %s;" % x for x in cls.imports), cls.name, n_anon + 1, sup_name, text)
            inner = parse_source(src, cls.path, keep_bodies=True)
            if inner:
                inner[0].outer_scope = cls
                for ic in inner:
                    if ic.outer is None:
                        ic.lexical_outer = cls
                idx.add(inner)
                shift = line_shift + line + body.count("\n", 0, off) - (2 + len(cls.imports))
                for ic in inner:
                    check_class(idx, ic, problems, shift)
        # ---- @Override
        if override:
            hit = None
            for a in ancestors:
                for mods, ptypes, _ in a.methods.get(name, []):
                    if len(ptypes) == len(params):
                        hit = (a, mods)
                        break
                if hit:
                    break
            if hit is None:
                bad(line, f"@Override {name}/{len(params)}: no ancestor declares it")
            else:
                a, mods = hit
                if "final" in mods or "static" in mods or "private" in mods:
                    bad(line, f"@Override {name}: {a.qual}.{name} is {'/'.join(sorted(mods & {'final', 'static', 'private'}))}")
                elif not accessible(mods, a, cls, True):
                    bad(line, f"@Override {name}: {a.qual}.{name} is not visible from package {cls.package}")
        # ---- declared names inside the body: parameters, locals, lambda parameters
        types = {p[1]: p[0] for p in params}
        for m in re.finditer(r"(?<![\w.])((?:final\s+)?[A-Z][\w.]*(?:<[^;(){}=]*?>)?(?:\[\])*|int|long|double|boolean|byte|short|char|float)(?:\[\])*\s+(\w+)\s*(?==|;|:|,)", body):
            types.setdefault(m.group(2), strip_generics(m.group(1)).replace("final", "").strip())
        for m in re.finditer(r"\(\s*(\w+(?:\s*,\s*\w+)*)\s*\)\s*->|(\w+)\s*->", body):
            for nm in re.findall(r"\w+", m.group(1) or m.group(2)):
                types.setdefault(nm, "?")
        for m in re.finditer(r"catch\s*\(\s*(?:final\s+)?[\w.|\s]+\s+(\w+)\s*\)", body):
            types.setdefault(m.group(1), "?")

        def type_of(name_):
            if name_ in types:
                return types[name_], None
            if name_ in cls.fields:
                return cls.fields[name_][1], None
            f = None
            for a in ancestors:
                if name_ in a.fields:
                    f = a
                    break
            if f is not None:
                return f.fields[name_][1], f
            return None, None

        # ---- super(...) constructor calls
        if kind == "ctor":
            for m in re.finditer(r"(?<![\w.])super\s*\(", body):
                nargs = count_args(body, m.end() - 1)
                sup = idx.resolve(cls.extends, cls) if cls.extends else None
                if sup is not None:
                    ok = [c for c in sup.ctors if arity_ok([c], nargs) and accessible(c[0], sup, cls, True)]
                    if not ok and (sup.ctors or nargs):
                        bad(line + body.count("\n", 0, m.start()), f"super(...) with {nargs} arguments: {sup.qual} has constructors of arity "
                            f"{sorted(len(c[1]) for c in sup.ctors)} (visible ones only count)")
        # ---- identifiers and member chains
        for m in re.finditer(r"(?<![\w.\"'])(new\s+)?([A-Za-z_]\w*)((?:\s*\.\s*[A-Za-z_]\w*)*)\s*(\()?", body):
            is_new, first, chain, call = m.group(1), m.group(2), m.group(3), m.group(4)
            ln = line + body.count("\n", 0, m.start())
            if first in KEYWORDS and first not in ("this", "super"):
                continue
            members = re.findall(r"[A-Za-z_]\w*", chain)
            if is_new:
                t = idx.resolve(".".join([first] + members), cls)
                if t is not None and call:
                    nargs = count_args(body, m.end() - 1)
                    if t.ctors and not [c for c in t.ctors if arity_ok([c], nargs) and accessible(c[0], t, cls, False)] and "abstract" not in t.mods:
                        bad(ln, f"new {t.name}(...) with {nargs} arguments: no visible constructor of that arity")
                continue
            cur, via_inh = None, False   # class of the expression so far
            rest = members
            if first in ("this", "super"):
                cur, via_inh = cls, True
            elif first in types or first in own_fields:
                tname, _ = type_of(first)
                cur = idx.resolve(tname, cls) if tname else None
                if cur is None:
                    continue
            else:
                inherited = None
                for a in ancestors:
                    if first in a.fields or (call and not members and first in a.methods):
                        inherited = a
                        break
                if call and not members:
                    if first in own_methods:
                        continue
                    if inherited is not None and first in inherited.methods:
                        mods = inherited.methods[first][0][0]
                        if not accessible(mods, inherited, cls, True):
                            bad(ln, f"{first}(...): {inherited.qual}.{first} is {visibility(mods)}, not visible from {cls.qual}")
                        elif not arity_ok(inherited.methods[first], count_args(body, m.end() - 1)):
                            bad(ln, f"{first}(...): no overload of {inherited.qual}.{first} takes {count_args(body, m.end() - 1)} arguments")
                        continue
                    encl = cls.outer if cls.outer is not None else getattr(cls, "lexical_outer", None)
                    outer_hit = encl is not None and find_member(idx, encl, first, "method")
                    if not outer_hit and first[0].islower() and not re.match(r"^(print|println)$", first):
                        bad(ln, f"{first}(...): no such method in {cls.name} or its ancestors")
                    continue
                if inherited is not None and first in inherited.fields:
                    mods, tname = inherited.fields[first]
                    if not accessible(mods, inherited, cls, True):
                        bad(ln, f"field {first}: {inherited.qual}.{first} is {visibility(mods)}, not visible from {cls.qual}")
                        continue
                    cur = idx.resolve(tname, inherited)
                    if cur is None:
                        continue
                else:
                    t = idx.resolve(first, cls)   # a class name: static access
                    k = 0
                    while t is not None and k < len(members) and idx.resolve(t.qual + "." + members[k], cls) is not None:
                        t = idx.resolve(t.qual + "." + members[k], cls)
                        k += 1
                    if t is None:
                        continue
                    cur, rest = t, members[k:]
            for k, name_ in enumerate(rest):
                last = k == len(rest) - 1
                want = "method" if (last and call) else "field"
                hit = find_member(idx, cur, name_, want)
                if hit is None:
                    if want == "field" and name_ == "length":
                        break
                    bad(ln, f"{cur.name}.{name_}{'(...)' if want == 'method' else ''}: no such {want} in {cur.qual} or its ancestors")
                    break
                owner, mods, tname, overloads = hit
                inh = via_inh or any(a.qual == owner.qual for a in ancestors) and (first in ("this", "super") or cur.qual == cls.qual)
                if not accessible(mods, owner, cls, inh):
                    bad(ln, f"{cur.name}.{name_}: {owner.qual}.{name_} is {visibility(mods)}, not visible from {cls.qual}")
                    break
                if overloads is not None and last and call and not arity_ok(overloads, count_args(body, m.end() - 1)):
                    bad(ln, f"{cur.name}.{name_}(...): no overload takes {count_args(body, m.end() - 1)} arguments")
                nxt = idx.resolve(tname, owner) if tname else None
                if nxt is None:
                    break
                cur, via_inh = nxt, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--glue", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "java", "src"))
    args = ap.parse_args()
    idx = Index()
    n_ref = 0
    for mod in ("core", "base"):
        for dp, _, files in os.walk(os.path.join(args.reference, mod, "src", "main", "java")):
            for f in files:
                if f.endswith(".java"):
                    idx.add(parse_file(os.path.join(dp, f)))
                    n_ref += 1
    if n_ref == 0:
        print("reference sources not found under", args.reference)
        return 2
    glue = []
    for dp, _, files in os.walk(args.glue):
        for f in sorted(files):
            if f.endswith(".java"):
                cs = parse_file(os.path.join(dp, f), keep_bodies=True)
                idx.add(cs)
                glue += cs
    problems = []
    for c in glue:
        check_class(idx, c, problems)
    for p in problems:
        print(p)
    print(f"checked {len(glue)} glue classes against {n_ref} reference files: {len(problems)} finding(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
