"""Random-parameter sweep against the UNMODIFIED reference (oracle/_ref) on the wave emulator -- developer tool, authoring container
only (needs /root/reference built into oracle/_ref).  usage: python tools/sweeps/sweep_align.py [cases] [seed]
Findings of round 2: loci that overflow the typical-case workspace, jumpRange on a path that overruns ref1."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, random, time
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_lib import RefLib
from manta_amd._capi import Lib, align_text
ref = RefLib(); emu = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so"))
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
def seq(n, alpha="ACGT"): return "".join(rng.choice(alpha) for _ in range(n))
def mutate(s):
    out=[]
    for c in s:
        x=rng.random()
        if x<0.03: out.append(rng.choice("ACGT"))
        elif x<0.04: continue
        elif x<0.05: out.append(c); out.append(rng.choice("ACGT"))
        elif x<0.055: out.append("N")
        else: out.append(c)
    return "".join(out)
bad=0; t0=time.time(); nerr=0
for it in range(N):
    kind = rng.choice([0,1,2])
    sc = [rng.choice([1,2,3]), rng.choice([-1,-4,-8]), rng.choice([0,-3,-12,-24]), rng.choice([0,-1,-2]), rng.choice([-1,-2,0]), rng.choice([0,0,1]) if kind!=2 else 0]
    extra = rng.choice([-100,-50,-20,-3,0])
    L = rng.randint(1, 260)
    base = seq(rng.randint(L, L+200))
    s = rng.randint(0, len(base)-L)
    q = mutate(base[s:s+L]) or "A"
    if rng.random()<0.3:  # large deletion / insertion
        p = rng.randint(0, len(q))
        q = q[:p] + seq(rng.randint(0,40)) + q[p+rng.randint(0,10):]
        q = q or "C"
    r1 = base if rng.random()<0.8 else seq(rng.randint(1,50))
    r2 = seq(rng.randint(1,150)) if kind==2 else None
    if kind==2 and rng.random()<0.6:
        r2 = seq(rng.randint(0,30)) + q[len(q)//2:] + seq(rng.randint(0,30))
    try:
        want = ref.align(kind, sc, extra, q, r1, r2)
    except Exception as e:
        nerr+=1; continue
    res = emu.align_batch(kind, sc, extra, [(q, r1) if kind!=2 else (q, r1, r2)], strict=False)[0]
    got = align_text(kind, res) if res["status"]==0 else "STATUS %d"%res["status"]
    if got != want:
        bad+=1
        if bad<=3: print("MISMATCH", it, kind, sc, extra, len(q), len(r1), len(r2 or "")); print(want); print(got)
print("align sweep", N, "mismatches", bad, "ref errors", nerr, "%.0fs"%(time.time()-t0))
