import re, sys
for l in open(sys.argv[1]):
    if " lib " not in l:
        print(l.strip()); continue
    name, rest = l.split("|")
    items = re.findall(r"(\d+)/s(\d+) ([\d.]+)", rest)
    best = sorted(items, key=lambda t: float(t[2]))[:3]
    bp = sorted([i for i in items if int(i[0]) >= 49], key=lambda t: float(t[2]))[:1]
    print(name.strip(), "| best:", ", ".join("%s/s%s %s" % b for b in best), "| best persistent:", ", ".join("%s/s%s %s" % b for b in bp))
