"""Parse the argument structs of include/neuray_hip.h and print the ctypes classes a binding needs.

    python tools/abi_structs.py            # the `class X(C.Structure)` source for every struct of the header

INTEGRATION.md Option B is this output (tests/test_c_abi.py re-parses the document's snippet against the header, so the
document cannot go stale against the ABI again)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'neuray_hip.h')

_KIND = {'int': 'C.c_int', 'float': 'C.c_float', 'size_t': 'C.c_size_t', 'unsigned': 'C.c_uint'}


def parse_structs(path=HEADER):
    """-> {struct name: [(field name, 'C.c_void_p' | 'C.c_int' | 'C.c_float' ...), ...]} in declaration order"""
    text = re.sub(r'/\*.*?\*/', '', open(path).read(), flags=re.S)
    out = {}
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            head, _, names = decl.rpartition(' ') if ',' not in decl else (None, None, None)
            if head is None:                                   # `int a, b, c`
                first, *rest = [p.strip() for p in decl.split(',')]
                head, _, n0 = first.rpartition(' ')
                names = [n0] + rest
            else:
                names = [names]
            base = head.replace('const ', '').strip()
            for n in names:
                ptr = '*' in base or n.startswith('*')
                kind = 'C.c_void_p' if ptr else _KIND[base.split()[0]]
                fields.append((n.lstrip('*'), kind))
        out[m.group(3)] = fields
    return out


def ctypes_source(structs=None, only=None):
    structs = structs or parse_structs()
    lines = []
    for name, fields in structs.items():
        if only and name not in only:
            continue
        lines.append('class %s(C.Structure):           # struct %s, include/neuray_hip.h' % (name, name))
        lines.append('    _fields_ = [')
        row = '        '
        for f, k in fields:
            item = "('%s', %s), " % (f, k)
            if len(row) + len(item) > 118:
                lines.append(row.rstrip())
                row = '        '
            row += item
        lines.append(row.rstrip())
        lines.append('    ]')
        lines.append('')
    return '\n'.join(lines)


if __name__ == '__main__':
    print(ctypes_source())
