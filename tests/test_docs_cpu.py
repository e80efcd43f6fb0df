"""Documents that name things the code defines: kept in step by the CPU suite."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_variable_the_library_reads_is_documented():
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "rdycore_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "rdycore_amd", "csrc", "*.hip")))
    read = set(re.findall(r'getenv\("(RDYHIP_[A-Z0-9_]+|ROCPROF_[A-Z0-9_]+)"\)', src))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(v for v in read if v not in doc)
    assert not missing, f"INTEGRATION.md (environment variables) does not mention {missing}"
    assert 10 <= len(read) <= 16          # measurement knobs, each read once at create (25, one of them per step)


def test_every_evidence_file_the_documents_cite_exists():
    missing = set()
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", "profiles/RESULTS_LOG.md", "profiles/README.md", "include/rdyhip.h"):
        txt = open(os.path.join(ROOT, doc)).read()
        names = set(re.findall(r"profiles/([A-Za-z0-9_.-]+\.(?:txt|json|csv|md|log))", txt))
        if doc.startswith("profiles/") or doc == "DESIGN.md":
            names |= set(re.findall(r"`(r0[1-9]_[A-Za-z0-9_.-]+\.(?:txt|json|csv|log))`", txt))
        missing |= {(doc, n) for n in names if not os.path.exists(os.path.join(ROOT, "profiles", n))}
    assert not missing, sorted(missing)
