"""Resolve every `#ifdef COTR_EXPERIMENTAL ... [#else ...] #endif` of a source file one way or the other (other conditionals are
left alone).  Used ONCE in round 5 to split the sources: the product files keep the -U side and have no research code in them, the
research library's forks (cotr_amd/csrc/experimental/<name>) keep the -D side.

    python tools/unifdef_exp.py {-D|-U} in_file out_file
"""
import re
import sys


def resolve(lines, defined):
    out, stack = [], []          # stack of ('exp', emitting_now) / ('other', None)
    for ln in lines:
        st = ln.strip()
        if re.match(r'#\s*ifdef\s+COTR_EXPERIMENTAL\b', st):
            stack.append(['exp', defined])
            continue
        if re.match(r'#\s*ifndef\s+COTR_EXPERIMENTAL\b', st):
            stack.append(['exp', not defined])
            continue
        if re.match(r'#\s*if', st):
            stack.append(['other', None])
        elif re.match(r'#\s*else\b', st) and stack and stack[-1][0] == 'exp':
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r'#\s*endif\b', st):
            top = stack.pop()
            if top[0] == 'exp':
                continue
        if all(f[1] for f in stack if f[0] == 'exp'):
            out.append(ln)
    assert not stack, 'unbalanced conditionals'
    return out


if __name__ == '__main__':
    mode, src, dst = sys.argv[1:4]
    res = resolve(open(src).read().splitlines(keepends=True), mode == '-D')
    open(dst, 'w').write(''.join(res))
