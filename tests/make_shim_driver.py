"""Generates the C driver of the host-side sanitizer test from include/dgtta.h: every entry point of the C ABI is called
WITHOUT a GPU - with all-zero arguments (NULL pointers, zero sizes), and the *_bytes size queries also with typical and
with very large dimensions - under AddressSanitizer + UndefinedBehaviorSanitizer.  The calls must be rejected by the
argument checks (negative return code, message set) before anything is launched."""
import re
import sys
from pathlib import Path

PROTO = re.compile(r"^(int|size_t|const char \*)\s*(dgtta_\w+)\(([^;]*?)\);", re.M | re.S)


def parse(header):
    text = re.sub(r"/\*.*?\*/", "", Path(header).read_text(), flags=re.S)
    out = []
    for ret, name, args in PROTO.findall(text):
        params = [a.strip() for a in " ".join(args.split()).split(",")] if args.strip() not in ("", "void") else []
        out.append((ret.strip(), name, params))
    return out


def kind(p):
    if "*" in p:
        return "ptr"
    t = p.rsplit(" ", 1)[0]
    if "float" in t or "double" in t:
        return "float"
    if "int64_t" in t or "uint64_t" in t or "size_t" in t:
        return "i64"
    return "int"


def call(name, params, val):
    vals = {"ptr": "0", "float": "0.0f", "i64": str(val.get("i64", 0)), "int": str(val.get("int", 0))}
    return f"{name}({', '.join(vals[kind(p)] for p in params)})"


def main(header, out):
    fns = parse(header)
    lines = ['#include <stdio.h>', '#include <string.h>', '#include "dgtta.h"', "int main(void) {", "  int bad = 0;"]
    for ret, name, params in fns:
        if ret == "const char *":
            lines.append(f'  printf("{name} -> %s\\n", {name}());')
        elif ret == "size_t":
            for tag, val in (("zero", {}), ("typical", {"int": 32, "i64": 2097152}), ("huge", {"int": 1 << 14, "i64": 1 << 40})):
                lines.append(f'  printf("{name} {tag} -> %zu\\n", {call(name, params, val)});')
        else:
            lines.append(f'  {{ int rc = {call(name, params, {})}; const char *m = dgtta_last_error();')
            lines.append(f'    printf("{name} -> %d: %s\\n", rc, m);')
            if params and name not in ("dgtta_reload_env",) and not name.endswith("_supported"):      # (predicates answer 0 / 1)
                lines.append(f'    if (rc >= 0 || !m || !m[0]) {{ printf("NOT REJECTED: {name}\\n"); bad = 1; }} }}')
            else:
                lines.append("  }")
    lines += ['  printf("functions %d\\n", ' + str(len(fns)) + ");", "  return bad;", "}"]
    Path(out).write_text("\n".join(lines) + "\n")
    return len(fns)


if __name__ == "__main__":
    print(main(sys.argv[1], sys.argv[2]))
