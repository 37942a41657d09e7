#!/usr/bin/env python3
"""Host time of ShardedLetkf._native_submit / _native_finish per top-level statement: the functions' source is re-compiled with a
time stamp in front of every top-level statement of their bodies (nothing in the shipped code).  python tools/submit_lines.py"""
import ast, collections, os, sys, textwrap, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd import sharded
from torch_assimilate_amd.sharded import ShardedLetkf

src_path = sharded.__file__
src = open(src_path).read()
lines = src.split("\n")
tree = ast.parse(src)
ACC = collections.defaultdict(lambda: [0, 0, 0])      # (function, line) -> [ns, calls, max ns]
_last = [None, 0]


def _pp(fn, ln):
    now = time.perf_counter_ns()
    if _last[0] is not None and _last[0][0] == fn:
        a = ACC[_last[0]]
        a[0] += now - _last[1]
        a[1] += 1
        a[2] = max(a[2], now - _last[1])
    _last[0], _last[1] = (fn, ln), time.perf_counter_ns()


def _end(fn):
    now = time.perf_counter_ns()
    if _last[0] is not None and _last[0][0] == fn:
        a = ACC[_last[0]]
        a[0] += now - _last[1]
        a[1] += 1
        a[2] = max(a[2], now - _last[1])
    _last[0] = None


def instrument(name):
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ShardedLetkf")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name)
    body = [s for s in fn.body if not (isinstance(s, ast.Expr) and isinstance(getattr(s, "value", None), ast.Constant))]
    out = lines[fn.lineno - 1:fn.end_lineno]
    off = fn.lineno
    ins = {}
    for s in body:
        ins[s.lineno] = "        _pp(%r, %d)" % (name, s.lineno)
    new = []
    for i, l in enumerate(out):
        ln = off + i
        if ln in ins:
            new.append(ins[ln])
        if l.strip().startswith("return") and l.startswith("        return"):
            new.append("        _end(%r)" % name)
        new.append(l)
    code = textwrap.dedent("\n".join(new))
    ns = dict(sharded.__dict__)
    ns["_pp"], ns["_end"] = _pp, _end
    exec(compile(code, src_path + ":" + name, "exec"), ns)
    setattr(ShardedLetkf, name, ns[name])


instrument("_submit_fast")
instrument("_native_submit")
instrument("_native_finish")
mia.build()
from torch_assimilate_amd import _cabi
for o in os.environ.get("OPTIONS", "").split():
    _cabi.set_option(o.split("=")[0], int(o.split("=")[1]))
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
depth = int(os.environ.get("DEPTH", "8"))
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=depth, copy_results=False)


def run(n):
    if os.environ.get("SERIAL"):
        for it in range(n):
            runner.assimilate(X, gx, ox, Yb, d)
        return
    pend = collections.deque()
    for it in range(n):
        pend.append(runner.submit(X, gx, ox, Yb, d))
        if len(pend) == depth:
            pend.popleft().result()
    while pend:
        pend.popleft().result()


run(300)
import gc
gc.collect(); gc.freeze()
run(300)
ACC.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 2000
run(N)
torch.cuda.synchronize()
print("instrumented loop: %.1f us/step" % (1e6 * (time.perf_counter() - t0) / N))
for fn in ("_submit_fast", "_native_submit", "_native_finish"):
    tot = sum(v[0] for (f, _), v in ACC.items() if f == fn)
    print("%s: %.1f us per step in total; statements over 0.25 us:" % (fn, tot / N / 1e3))
    for (f, ln), v in sorted(ACC.items(), key=lambda kv: kv[0][1]):
        if f == fn and v[0] / N / 1e3 >= 0.25:
            print("  line %4d  %6.2f us  (max %7.1f)  x%5d   %s" % (ln, v[0] / N / 1e3, v[2] / 1e3, v[1], lines[ln - 1].strip()[:100]))
