"""The documents cite measurements by the directory under profiles/ that holds them: every directory DESIGN.md, README.md
and INTEGRATION.md name exists (or is one of the name patterns `rd5*` / `rd6*`), and profiles/MANIFEST.md has a row for
every directory of rounds 5 and 6."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dirs():
    return {d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.isdir(os.path.join(ROOT, "profiles", d))}


def test_cited_profile_directories_exist():
    have = _dirs()
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        cited = set(re.findall(r"profiles/(r[0-9a-z]+)", text)) | set(re.findall(r"`(rd?[0-9][0-9a-z]*)`", text)) | set(re.findall(r"`(rd?[0-9][0-9a-z]*)/", text))
        missing = sorted(c for c in cited if c not in have and c not in ("rd5", "rd6", "r4", "r5", "r3"))
        assert not missing, "%s cites profiles that are not in the tree: %s" % (doc, missing)


def test_manifest_lists_the_directories_of_rounds_5_and_6():
    manifest = open(os.path.join(ROOT, "profiles", "MANIFEST.md")).read()
    missing = sorted(d for d in _dirs() if d.startswith(("rd5", "rd6")) and not re.search(r"\b%s\b" % re.escape(d), manifest))
    assert not missing, "profiles/MANIFEST.md has no row for: %s" % missing
