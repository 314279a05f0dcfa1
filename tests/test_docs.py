"""The design document stays one ledger: under 60 KB, every dropped experiment's patch named in it, the per-round
narratives under docs/history/.  CPU."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_is_one_ledger_under_60_kb():
    design = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    assert len(design.encode("utf-8")) < 60_000
    for name in sorted(os.listdir(os.path.join(ROOT, "scripts", "dropped"))):
        assert name in design, "scripts/dropped/%s has no row in DESIGN.md section 7" % name
    assert os.path.exists(os.path.join(ROOT, "docs", "history", "DESIGN_rounds1-5.md"))
    for section in ("## 1. The path and its boundary", "## 2. Oracle", "## 3. Data layout in HBM", "## 4. Kernels", "## 5. Measurement", "## 6. Multi-GPU",
                    "## 7. Ledger", "## 9. Out of scope"):
        assert section in design, section
    assert "parity unpinned" in design
