"""A small `unifdef`: resolve the preprocessor conditionals of a source file for macros whose value is being FIXED, leaving
every other conditional untouched (round 6 prune: measured-negative experiment branches leave the shipped translation units).

usage: unifdef.py file NAME=value [NAME=value ...] [--drop-defs]
  * `#if EXPR` / `#elif EXPR` / `#else` / `#endif` chains whose conditions contain only fixed names (and integer literals, ! && ||
    == != < > <= >= & | + - parentheses, `defined(NAME)`) are resolved: the live branch's body stays, the directives and the dead
    branches go.  A chain with any condition that mentions an unknown name is left as it is (its body is still processed).
  * --drop-defs: `#ifndef NAME` / `#define NAME ...` / `#endif` default-definition triplets of fixed names are removed as well
    (do that only when no plain-code use of NAME is left - the script lists the remaining uses).
The result is written back to the file; the number of lines removed is printed."""
import re
import sys


def known(expr, env):
    names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
    return all(n in env for n in names)


def evaluate(expr, env):
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in env else "0", expr)
    e = re.sub(r"[A-Za-z_]\w*", lambda m: str(env[m.group(0)]), e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e, {"__builtins__": {}}, {}))


def process(lines, env, drop_defs):
    out, i, n = [], 0, len(lines)

    def chain_at(start):
        """-> list of (directive kind, expr, body_start, body_end) and the index after #endif; bodies are nested-aware"""
        parts, depth, j = [], 0, start
        kind, expr = "if", lines[start].split(None, 1)[1].split("//")[0].strip() if lines[start].lstrip().startswith("#if ") else None
        body0 = start + 1
        j = start + 1
        while j < n:
            s = lines[j].lstrip()
            if s.startswith(("#if ", "#ifdef", "#ifndef")):
                depth += 1
            elif s.startswith("#endif"):
                if depth == 0:
                    parts.append((kind, expr, body0, j))
                    return parts, j + 1
                depth -= 1
            elif depth == 0 and s.startswith("#elif"):
                parts.append((kind, expr, body0, j))
                kind, expr, body0 = "elif", s.split(None, 1)[1].split("//")[0].strip(), j + 1
            elif depth == 0 and s.startswith("#else"):
                parts.append((kind, expr, body0, j))
                kind, expr, body0 = "else", None, j + 1
            j += 1
        raise SystemExit("unterminated conditional at line %d" % (start + 1))

    while i < n:
        s = lines[i].lstrip()
        if drop_defs and s.startswith("#ifndef") and i + 2 < n:
            name = s.split()[1]
            if name in env and lines[i + 1].lstrip().startswith("#define " + name):
                j = i + 2
                while j < n and not lines[j].lstrip().startswith("#endif"):        # continuation / comment lines of the definition
                    j += 1
                if j < n and not any(l.lstrip().startswith(("#if", "#el")) for l in lines[i + 2:j]):
                    i = j + 1
                    continue
        if s.startswith("#if "):
            parts, after = chain_at(i)
            conds = [p for p in parts if p[0] != "else"]
            if all(known(p[1], env) for p in conds):
                live = None
                for kind, expr, b0, b1 in parts:
                    if kind == "else" or evaluate(expr, env):
                        live = (b0, b1)
                        break
                if live is not None:
                    out.extend(process(lines[live[0]:live[1]], env, drop_defs))
                i = after
                continue
            # unknown: keep the directives, process the bodies
            pos = i
            for kind, expr, b0, b1 in parts:
                out.extend(lines[pos:b0])
                out.extend(process(lines[b0:b1], env, drop_defs))
                pos = b1
            out.extend(lines[pos:after])
            i = after
            continue
        out.append(lines[i])
        i += 1
    return out


def main():
    path = sys.argv[1]
    drop = "--drop-defs" in sys.argv
    env = {}
    for a in sys.argv[2:]:
        if "=" in a:
            k, v = a.split("=", 1)
            env[k] = int(v)
    lines = open(path).read().split("\n")
    res = process(lines, env, drop)
    open(path, "w").write("\n".join(res))
    print("%s: %d -> %d lines" % (path, len(lines), len(res)))
    text = "\n".join(res)
    for k in env:
        uses = [m.start() for m in re.finditer(r"\b%s\b" % k, text)]
        if uses:
            print("   %s: %d plain uses left" % (k, len(uses)))


if __name__ == "__main__":
    main()
