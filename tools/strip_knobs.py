#!/usr/bin/env python3
"""Resolve preprocessor conditionals on a given set of macros as UNDEFINED (a small `unifdef -U`): the experiment / ablation
knobs of the fused kernels leave the product sources this way (the reverse patch is kept under tools/experiments/).
usage: strip_knobs.py file MACRO [MACRO ...]     (rewrites the file in place)"""
import re, sys

def evaluate(expr, undef):
    """True/False if the condition only involves macros of `undef` (all undefined), else None (leave the directive alone)."""
    names = set(re.findall(r"[A-Za-z_]\w*", re.sub(r"/\*.*?\*/", "", expr))) - {"defined"}
    if not names or not names <= undef:
        return None
    e = re.sub(r"/\*.*?\*/", "", expr)
    e = re.sub(r"defined\s*\(\s*\w+\s*\)|defined\s+\w+", "0", e)
    e = re.sub(r"[A-Za-z_]\w*", "0", e)            # an undefined macro in #if arithmetic is 0
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    e = e.replace(" not =", "!=")
    return bool(eval(e))

def strip(lines, undef):
    out, stack = [], []    # stack entries: dict(kind='resolved'|'kept', taken=bool, emitting=bool, parent_emit=bool)
    emit = lambda: all(s["emitting"] for s in stack)
    for ln in lines:
        m = re.match(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", ln.rstrip("\n"))
        if not m:
            if emit():
                out.append(ln)
            continue
        d, rest = m.group(1), m.group(2).strip()
        if d in ("ifdef", "ifndef", "if"):
            if d == "ifdef":
                name = rest.split()[0]
                val = False if name in undef else None
            elif d == "ifndef":
                name = rest.split()[0]
                val = True if name in undef else None
            else:
                val = evaluate(rest, undef)
            if val is None:
                if emit():
                    out.append(ln)
                stack.append({"kind": "kept", "emitting": True})
            else:
                stack.append({"kind": "resolved", "emitting": val, "taken": val, "converted": False})
        elif d == "elif":
            s = stack[-1]
            if s["kind"] == "kept":
                if emit():
                    out.append(ln)
            else:
                if s["taken"]:
                    s["emitting"] = False
                else:
                    val = evaluate(rest, undef)
                    if val is None:          # the chain continues on a condition we keep: it becomes the chain's #if
                        s["emitting"] = True
                        outer = all(x["emitting"] for x in stack[:-1])
                        if outer:
                            out.append(re.sub(r"#\s*elif", "#if", ln, count=1))
                        s["kind"], s["converted"] = "kept", True
                    else:
                        s["emitting"], s["taken"] = val, val
        elif d == "else":
            s = stack[-1]
            if s["kind"] == "kept":
                if emit():
                    out.append(ln)
            else:
                s["emitting"] = not s["taken"]
                s["taken"] = True
        else:  # endif
            s = stack.pop()
            if s["kind"] == "kept" and all(x["emitting"] for x in stack):
                out.append(ln)
    assert not stack
    return out

if __name__ == "__main__":
    path, undef = sys.argv[1], set(sys.argv[2:])
    src = open(path).read().split("\n")
    # join continuation lines of directives is not needed here: the knob directives are single lines
    res = strip([l + "\n" for l in src[:-1]] + ([src[-1]] if src[-1] else []), undef)
    open(path, "w").write("".join(res))
