"""Guards against stale documentation: the entry-point count DESIGN.md quotes is the header's, every profiles/ file the docs cite
exists, every golden file has a generator script that names it, and every C-ABI entry is cited in INTEGRATION.md or DESIGN.md."""
import os
import re

from snerf_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
read = lambda *p: open(os.path.join(REPO, *p)).read()


def test_entry_point_count_matches_the_header():
    n = len(_lib.parse_header())
    m = re.search(r"C-ABI\*\* of (\d+) entry points", read("DESIGN.md"))
    assert m and int(m.group(1)) == n, (m and m.group(1), n)
    assert len(_lib.parse_header(_lib.IO_HEADER_PATH)) == 3


def test_cited_profiles_exist():
    have = set(os.listdir(os.path.join(REPO, "profiles")))
    for doc in ("DESIGN.md", "profiles/README.md", "README.md", "bench.py"):
        for name in re.findall(r"profiles/(r[1-6]_[a-z]+_[A-Za-z0-9_.]+?\.(?:txt|log))", read(*doc.split("/"))):
            assert name in have, (doc, name)
    for name in re.findall(r"`(r[1-6]_[a-z]+_[A-Za-z0-9_.]+?\.(?:txt|log))`", read("profiles", "README.md")):
        assert name in have, name


def test_every_golden_file_has_a_generator():
    gens = "".join(read("oracle", f) for f in os.listdir(os.path.join(REPO, "oracle")) if f.startswith("gen_golden"))
    for f in os.listdir(os.path.join(REPO, "tests", "golden")):
        stem = f[:-4]
        assert stem in gens or re.sub(r"_[a-z0-9]+$", "", stem) in gens, f


def test_every_entry_point_is_documented():
    docs = read("DESIGN.md") + read("INTEGRATION.md") + read("snerf_amd", "ops.py") + read("snerf_amd", "foreground.py")
    for name in _lib.parse_header():
        if name != "snerf_version":
            assert name in docs, name
