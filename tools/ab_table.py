"""Tabulate tools/scripts/multi_ab.sh output: python tools/ab_table.py <file> <lib> [<lib> ...]"""
import collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 3:
        try: d[p[1]][p[0]].append(float(p[2]))
        except ValueError: pass
libs = sys.argv[2:]
print("layer", *libs)
for k, v in d.items():
    print(k, *["/".join("%.3f" % x for x in v[l]) for l in libs], *["%+.1f %%" % (100 * (min(v[l]) / min(v[libs[0]]) - 1)) for l in libs[1:] if v[l] and v[libs[0]]])
