"""The documents point at files; the files must exist (no GPU).  STATUS.md / DESIGN.md / README.md / INTEGRATION.md and the two
indexes name profiles, tests, tools and sources as evidence -- a pointer that dangles is a claim nobody can check."""
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
DOCS = ["STATUS.md", "DESIGN.md", "DESIGN_HISTORY.md", "README.md", "INTEGRATION.md", "profiles/README.md", "tools/README.md"]
PATH = re.compile(r"(?<![A-Za-z0-9_/])((?:profiles|tests|tools|nbody_amd|include|oracle)/[A-Za-z0-9_./-]+\.(?:txt|json|csv|py|sh|hip|c|h|md|err))")
BARE_PROFILE = re.compile(r"(?<![A-Za-z0-9_/])(r0[1-9]_[A-Za-z0-9_.-]+\.(?:txt|json|csv|err))")
# named on purpose although gone: the index says so in the same sentence
KNOWN_GONE = {"tools/gen_body.py"}


@pytest.mark.parametrize("doc", DOCS)
def test_every_file_a_document_points_at_exists(doc):
    text = open(os.path.join(ROOT, doc)).read()
    missing = set()
    for m in PATH.finditer(text):
        p = m.group(1).rstrip(".")
        if "{" in text[m.start():m.end() + 1] or p in KNOWN_GONE or "ubenchN" in p:
            continue
        if not os.path.exists(os.path.join(ROOT, p)):
            missing.add(p)
    for m in BARE_PROFILE.finditer(text):       # `r05_persist_probe.txt` with the directory left out
        before = text[max(0, m.start() - 1):m.start()]
        p = m.group(1).rstrip(".")
        if before == "/" or "{" in p:
            continue
        if not os.path.exists(os.path.join(ROOT, "profiles", p)):
            missing.add("profiles/" + p)
    assert not missing, f"{doc} points at files that do not exist: {sorted(missing)}"


def test_status_page_stays_one_page():
    assert len(open(os.path.join(ROOT, "STATUS.md")).read().splitlines()) <= 80


@pytest.mark.parametrize("doc", ["STATUS.md", "DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "include/nbody_hip.h",
                                 "nbody_amd/csrc/nbody_hip_tuning.h"])
def test_every_test_a_document_names_exists(doc):
    """`test_...` names are cited as evidence: each must be a test function of this suite (or an unambiguous prefix of one, for the
    names the documents shorten), a test module, or the reference's own test file."""
    import glob
    names, modules = set(), {"test_particle_sort"}          # the reference's test/test_particle_sort.c
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        modules.add(os.path.basename(f)[:-3])
        names |= set(re.findall(r"^def (test_[A-Za-z0-9_]+)", open(f).read(), re.M))
    text = open(os.path.join(ROOT, doc)).read()
    unknown = {t for t in re.findall(r"(?<![A-Za-z0-9_])(test_[a-z0-9_]+)", text)
               if t not in modules and t.rstrip("_") not in modules and not any(n.startswith(t.rstrip("_")) for n in names)}
    assert not unknown, f"{doc} cites tests that do not exist: {sorted(unknown)}"


def test_status_headline_table_is_the_drivers_own_record():
    """STATUS.md's per-round headline rows are copied from the driver's BENCH_rNN.json files: each listed round must match its
    record (ms/step to 0.1, fraction to 0.001) -- the page quotes the driver, not the builder."""
    import json
    text = open(os.path.join(ROOT, "STATUS.md")).read()
    rows = re.findall(r"^\| r(\d) \| ([0-9.]+) \| ([0-9.]+)·10\^12 \| ([0-9.]+) \|$", text, re.M)
    assert len(rows) >= 5
    for rnd, ms, rate, frac in rows:
        path = os.path.join(ROOT, f"BENCH_r0{rnd}.json")
        if not os.path.exists(path):
            pytest.skip(f"{path} not in this checkout")
        line = json.load(open(path))["parsed"]
        assert abs(line["ms_per_step"] - float(ms)) < 0.06, (rnd, line["ms_per_step"], ms)
        assert abs(line["value"] / 1e12 - float(rate)) < 0.006, (rnd, line["value"], rate)
        assert abs(line["roofline"]["frac"] - float(frac)) < 0.0006, (rnd, line["roofline"]["frac"], frac)
