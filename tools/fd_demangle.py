"""rocprofv3 leaves kernel names with _Float16 template arguments mangled (its demangler does not know DF16_).  This restates the few
Itanium-ABI productions our kernel symbols use -- _Z <len><name> I <template args> E <parameter types> -- and returns `name<args>`."""
import re


def _arg(s, i):
    if s.startswith("DF16_", i): return "_Float16", i + 5
    if s.startswith("DF16b", i): return "__bf16", i + 5
    if s[i] == "f": return "float", i + 1
    if s[i] == "d": return "double", i + 1
    if s[i] == "i": return "int", i + 1
    if s[i] == "b": return "bool", i + 1
    if s[i] == "L":                                   # literal: L <type> <value> E
        j = s.index("E", i)
        ty, val = s[i + 1], s[i + 2:j]
        if val.startswith("n"): val = "-" + val[1:]
        if ty == "b": val = "true" if val == "1" else "false"
        return val, j + 1
    m = re.match(r"(\d+)", s[i:])
    if m:                                             # <len><identifier>
        n = int(m.group(1)); k = i + len(m.group(1))
        return s[k:k + n], k + n
    raise ValueError(s[i:])


def demangle(name):
    name = name.strip()
    if not name.startswith("_Z"):
        return name.split("(")[0].replace("void ", "").strip()
    try:
        m = re.match(r"_Z(\d+)", name)
        n = int(m.group(1)); k = 2 + len(m.group(1))
        ident, i = name[k:k + n], k + n
        if i >= len(name) or name[i] != "I":
            return ident
        i += 1
        args = []
        while name[i] != "E":
            a, i = _arg(name, i)
            args.append(a)
        return "%s<%s>" % (ident, ", ".join(args))
    except Exception:
        return name


if __name__ == "__main__":
    import sys
    for a in sys.argv[1:]:
        print(demangle(a))
