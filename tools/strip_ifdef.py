"""Remove every `#ifdef SYMBOL ... [#else ...] #endif` block from C / C++ sources as if SYMBOL were undefined (the #else branch stays).
Nested conditionals inside the removed / kept parts are tracked.  `python tools/strip_ifdef.py PM_EXPERIMENTS file...` rewrites the
files in place -- how round 5 took the experiment variants out of the product sources (experiments/README.md)."""
import re
import sys


def strip(text, sym):
    out, stack = [], []      # stack entries: ('ours', keeping_now) or ('other', None)
    for line in text.split('\n'):
        s = line.strip()
        m_if = re.match(r'#\s*(ifdef|ifndef|if)\b(.*)', s)
        if m_if:
            kind, rest = m_if.group(1), m_if.group(2)
            name = rest.split('//')[0].strip()
            if kind in ('ifdef', 'ifndef') and name == sym:
                stack.append(['ours', kind == 'ifndef'])
                continue
            stack.append(['other', None])
        elif re.match(r'#\s*else\b', s) and stack and stack[-1][0] == 'ours':
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r'#\s*endif\b', s):
            top = stack.pop() if stack else ['other', None]
            if top[0] == 'ours':
                continue
        if all(k != 'ours' or keep for k, keep in stack):
            out.append(line)
    assert not stack, 'unbalanced conditionals'
    return '\n'.join(out)


if __name__ == '__main__':
    sym = sys.argv[1]
    for path in sys.argv[2:]:
        src = open(path).read()
        new = strip(src, sym)
        if new != src:
            open(path, 'w').write(new)
            print('stripped', path, src.count('\n') - new.count('\n'), 'lines')
