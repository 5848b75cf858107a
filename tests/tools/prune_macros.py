"""Resolve preprocessor conditionals on macros that are never defined any more (round 5: the losing experiments leave the hot files; they live on
under the git tag r04-experiments). usage: prune_macros.py file MACRO [MACRO ...] -- rewrites the file in place."""
import re
import sys


def prune(text, dead):
    out = []
    # stack entries: [kind, state] kind = 'keep' (conditional left in the file) | 'res' (resolved); state for 'res': 'taking' | 'skipping' | 'done'
    stack = []

    def emitting():
        return all(not (k == "res" and st != "taking") for k, st in stack)

    def ev(expr):
        e = expr.split("//")[0].strip()
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "0" if m.group(1) in dead else m.group(0), e)
        if re.fullmatch(r"[01!&|() ]+", e):
            return bool(eval(e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")))
        if re.match(r"^0\s*&&", e):
            return False
        return None

    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2).strip()
        if d in ("ifdef", "ifndef", "if"):
            if d == "if":
                v = ev(rest)
            else:
                name = rest.split()[0]
                v = (d == "ifndef") if name in dead else None
            if v is None:
                if emitting():
                    out.append(line)
                stack.append(["keep", None])
            else:
                stack.append(["res", "taking" if v else "skipping"])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "keep":
                if emitting():
                    out.append(line)
            else:
                if top[1] == "taking":
                    top[1] = "done"
                elif top[1] == "skipping":
                    v = ev(rest)
                    if v is None:
                        raise SystemExit("cannot resolve #elif %s after a dead branch" % rest)
                    top[1] = "taking" if v else "skipping"
        elif d == "else":
            top = stack[-1]
            if top[0] == "keep":
                if emitting():
                    out.append(line)
            else:
                top[1] = "taking" if top[1] == "skipping" else "done"
        else:
            top = stack.pop()
            if top[0] == "keep" and emitting():
                out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    path, dead = sys.argv[1], set(sys.argv[2:])
    src = open(path).read()
    open(path, "w").write(prune(src, dead))
