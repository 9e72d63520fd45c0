#!/usr/bin/env python3
"""usage: devn_convert.py file.hip kernel:param [kernel:param ...]
Source transformation used while teaching kernels to take their length from HBM: parameter `int64_t <param>` of
__global__ function <kernel> becomes `DevN <param>_d` and the body starts with `const int64_t <param> = dev_n(<param>_d);`
(cfx_common.h: DevN converts from a host int64_t, so exact call sites compile unchanged)."""
import re, sys

def convert(src, kernel, param):
    m = re.search(r'__global__[^;{]*?\b' + re.escape(kernel) + r'\s*\(', src)
    if not m:
        raise SystemExit(f"kernel {kernel} not found")
    i = m.end()
    depth, j = 1, i
    while depth:
        c = src[j]
        depth += c == '('
        depth -= c == ')'
        j += 1
    params = src[i:j - 1]
    pat = re.compile(r'\bint64_t\s+' + re.escape(param) + r'\b')
    if not pat.search(params):
        raise SystemExit(f"{kernel}: parameter int64_t {param} not found")
    params2 = pat.sub(f'DevN {param}_d', params, count=1)
    k = src.index('{', j)
    body_insert = f"\n  const int64_t {param} = dev_n({param}_d);"
    return src[:i] + params2 + src[j - 1:k + 1] + body_insert + src[k + 1:]

def main():
    path = sys.argv[1]
    src = open(path).read()
    for spec in sys.argv[2:]:
        kernel, param = spec.split(':')
        src = convert(src, kernel, param)
    open(path, 'w').write(src)

if __name__ == '__main__':
    main()
