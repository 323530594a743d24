#!/usr/bin/env python3
"""Generates tests/golden/*.json from the reference's OWN unit tests (run in the authoring container only).

For every known-answer test that pins the hot path this script
  1. parses the inputs and the asserted expectations straight out of the reference test sources under
     /root/reference/src/c++/lib/{alignment,assembly}/test/*.cpp (nothing is copied: only the test vectors, i.e.
     sequences, option values and expected strings/numbers, are extracted), and
  2. runs the UNMODIFIED reference implementation (oracle/_ref/libmanta_ref.so) on those inputs and stores its full
     canonical-text output (oracle/FORMAT.md) next to the asserted expectations.

The committed JSON travels to the GPU box; /root/reference does not.  tests/test_golden.py checks that
  (a) the stored reference output satisfies every expectation the reference test asserts (pins the _ref build),
  (b) the CPU restatement reproduces the stored reference output exactly (pins the oracle),
  (c) the HIP path reproduces it exactly (`-m gpu`, and on the wave emulator in the CPU tier).
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import RefLib, asm_opts, ASM_DEFAULTS  # noqa: E402

REF = os.environ.get("MANTA_REFERENCE", "/root/reference")
LIB = os.path.join(REF, "src/c++/lib")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def split_cases(src):
    """yield (name, body) for every BOOST_AUTO_TEST_CASE"""
    for m in re.finditer(r"BOOST_AUTO_TEST_CASE\((\w+)\)\s*\{", src):
        depth, i = 1, m.end()
        while depth:
            c = src[i]
            depth += (c == "{") - (c == "}")
            i += 1
        yield m.group(1), src[m.end():i - 1], src[:m.start()].count("\n") + 1


def helper_scores(src):
    """helper function name -> (scores list of 6 [offEdge/allowEdge may be parameter names], extra)"""
    out = {}
    for m in re.finditer(r"static\s+\w+<score_t>\s+(\w+)\s*\(([^)]*)\)\s*\{", src):
        depth, i = 1, m.end()
        while depth:
            c = src[i]
            depth += (c == "{") - (c == "}")
            i += 1
        body = src[m.end():i]
        sm = re.search(r"scores\(([^)]*)\)", body)
        sc = [s.strip() for s in sm.group(1).split(",")]
        extra = None
        em = re.search(r"jumpScore\((-?\d+)\)", body) or re.search(r"aligner\(scores,\s*(-?\d+)\)", body)
        if em:
            extra = int(em.group(1))
        defaults = {}
        for pm in re.finditer(r"const\s+\w+\s+(\w+)\s*=\s*(-?\w+)", m.group(2)):
            defaults[pm.group(1)] = pm.group(2)
        params = [p.split("=")[0].strip().split()[-1] for p in m.group(2).split(",")]
        params = [re.sub(r"[&*]", "", p) for p in params]
        out[m.group(1)] = (sc, extra, params, defaults)
    return out


def to_int(tok, env):
    tok = env.get(tok, tok)
    if tok in ("true", "false"):
        return int(tok == "true")
    return int(tok)


def parse_aligner_tests(path, kind):
    src = strip_comments(open(path).read())
    helpers = helper_scores(src)
    cases = []
    for name, body, line in split_cases(src):
        consts = dict(re.findall(r"std::string\s+(\w+)\(\s*\"([^\"]*)\"\s*\)", body))
        for rm in re.finditer(r"(\w+)\s*=\s*(testAlign\w*)\(([^)]*)\)", body):
            rvar, helper, args = rm.group(1), rm.group(2), [a.strip() for a in rm.group(3).split(",")]
            sc, extra, params, defaults = helpers[helper]
            env = dict(defaults)
            for p, a in zip(params, args):
                env[p] = a
            scores = [to_int(s, env) for s in sc]
            while len(scores) < 6:
                scores.append(0)
            seqs = [consts[a] for a in args if a in consts]
            expect = {}
            for em in re.finditer(r"BOOST_REQUIRE_EQUAL\(\s*(.+?),\s*([^,;]+?)\s*\);", body):
                lhs, rhs = em.group(1).strip(), em.group(2).strip().rstrip("u")
                if not re.search(r"\b%s\b" % rvar, lhs):
                    continue
                key = None
                m2 = re.match(r"apath_to_cigar\(%s\.align(\d?)\.apath\)" % rvar, lhs)
                if m2:
                    key = "cigar" + (m2.group(1) or "1")
                m2 = m2 or re.match(r"%s\.align(\d?)\.beginPos" % rvar, lhs)
                if key is None and m2:
                    key = "begin" + (m2.group(1) or "1")
                if key is None:
                    for field in ("score", "jumpInsertSize", "jumpRange", "isJumped"):
                        if lhs == "%s.%s" % (rvar, field):
                            key = field
                if key is None:
                    raise SystemExit("unparsed expectation in %s: %s" % (name, lhs))
                expect[key] = rhs.strip('"') if rhs.startswith('"') else int(rhs)
            cases.append(dict(source="%s:%d" % (os.path.relpath(path, REF), line), name=name + ("/" + rvar if rvar != "result" else ""),
                              kind=kind, scores=scores, extra=extra if extra is not None else 0, query=seqs[0], ref1=seqs[1],
                              ref2=seqs[2] if len(seqs) > 2 else None, expect=expect))
    return cases


def parse_assembler_tests(path):
    src = strip_comments(open(path).read())
    cases = []
    for name, body, line in split_cases(src):
        if "runIterativeAssembler" not in body:
            continue
        opts = dict(ASM_DEFAULTS)
        for om in re.finditer(r"assembleOpt\.(\w+)\s*=\s*(\d+)\s*;", body):
            opts[om.group(1)] = int(om.group(2))
        reads = re.findall(r"reads\.emplace_back\(\s*\"([^\"]*)\"\s*\)", body)
        # expand the simple `for (unsigned i(0); i < N; ++i) { ... [i] ... }` loops of these tests
        flat = body
        for lm in re.finditer(r"for\s*\(unsigned\s+i\(0\);\s*i\s*<\s*(\d+);\s*\+\+i\)\s*\{(.*?)\}", body, flags=re.S):
            flat = flat.replace(lm.group(0), "".join(lm.group(2).replace("[i]", "[%d]" % i) for i in range(int(lm.group(1)))))
        expect = {}
        for em in re.finditer(r"BOOST_REQUIRE_EQUAL\(\s*(.+?),\s*([^;]+?)\s*\);", flat):
            lhs, rhs = em.group(1).strip(), em.group(2).strip()
            expect[lhs] = rhs.strip('"') if rhs.startswith('"') else int(rhs.rstrip("u"))
        for em in re.finditer(r"BOOST_REQUIRE\(\s*(!?)\s*(readInfo\[\d+\]\.isUsed)\s*\);", flat):
            expect[em.group(2)] = 0 if em.group(1) else 1
        cases.append(dict(source="%s:%d" % (os.path.relpath(path, REF), line), name=name, opts=opts, reads=reads, expect=expect))
    return cases


def main():
    ref = RefLib()
    align_cases = []
    align_cases += parse_aligner_tests(os.path.join(LIB, "alignment/test/GlobalAlignerTest.cpp"), 0)
    align_cases += parse_aligner_tests(os.path.join(LIB, "alignment/test/GlobalLargeIndelAlignerTest.cpp"), 1)
    align_cases += parse_aligner_tests(os.path.join(LIB, "alignment/test/GlobalJumpAlignerTest.cpp"), 2)
    for c in align_cases:
        c["ref_text"] = ref.align(c["kind"], c["scores"], c["extra"], c["query"], c["ref1"], c["ref2"])
    json.dump(align_cases, open(os.path.join(HERE, "aligner_reference_tests.json"), "w"), indent=1)
    print("aligner cases:", len(align_cases))

    asm_cases = parse_assembler_tests(os.path.join(LIB, "assembly/test/IterativeAssemblerTest.cpp"))
    for c in asm_cases:
        c["ref_text"] = ref.assemble(asm_opts(**c["opts"]), c["reads"])
    json.dump(asm_cases, open(os.path.join(HERE, "assembler_reference_tests.json"), "w"), indent=1)
    print("assembler cases:", len(asm_cases))


if __name__ == "__main__":
    main()
