#!/usr/bin/env python3
"""Re-wrap the paragraphs and list items of markdown files to <= WIDTH bytes (tables, headings, code fences kept)."""
import re, sys, textwrap
WIDTH = 118
def reflow(text):
    out, para, fence = [], [], False
    def flush():
        if not para: return
        first = para[0]
        m = re.match(r'^(\s*)([-*] |\d+\. )?', first)
        ind = m.group(1) + (' ' * len(m.group(2)) if m.group(2) else '')
        body = ' '.join([first.strip()] + [p.strip() for p in para[1:]])
        cur = m.group(1)                           # greedy wrap on the UTF-8 byte length (what `awk length` counts)
        for word in body.split():
            cand = cur + ('' if cur.strip() == '' else ' ') + word
            if len(cand.encode()) > WIDTH and cur.strip():
                out.append(cur); cand = ind + word
            cur = cand
        out.append(cur)
        para.clear()
    for ln in text.split('\n'):
        if ln.startswith('```'):
            flush(); fence = not fence; out.append(ln); continue
        if fence or ln.startswith('|') or ln.startswith('#') or not ln.strip():
            flush(); out.append(ln); continue
        if re.match(r'^\s*([-*] |\d+\. )', ln):
            flush()
        para.append(ln)
    flush()
    return '\n'.join(out)
for f in sys.argv[1:]:
    t = open(f).read(); r = reflow(t)
    if r != t: open(f, 'w').write(r)
    bad = [i + 1 for i, l in enumerate(r.split('\n')) if len(l.encode()) > 120 and not l.startswith('|')]
    print(f, len(r.split('\n')), 'lines; over 120 (non-table):', bad)
