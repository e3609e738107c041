"""tools/ab_variants.py (the A/B table bench.py carries as `extra.ab_variants`) end to end on the simulated device: every experiment's
build variant is compiled for the CPU with the same -D flags, every child runs at a small size, and every variant reproduces the base
library's results bit for bit (visible ids, sorted pairs + carried lod / Pose::frame state, palettes and skinned positions)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _hostsim_lib(flags):
    from tests.hostsim import build as hostsim_build

    old = os.environ.get("LMX_HOSTSIM_EXTRA")
    os.environ["LMX_HOSTSIM_EXTRA"] = " ".join(flags)
    try:
        return hostsim_build.build()
    finally:
        if old is None:
            del os.environ["LMX_HOSTSIM_EXTRA"]
        else:
            os.environ["LMX_HOSTSIM_EXTRA"] = old


def test_ab_table_on_the_simulated_device(monkeypatch):
    import ab_variants as AB
    from tests.hostsim import build as hostsim_build

    base = hostsim_build.build()
    libs = {name: _hostsim_lib(v[2]) for name, v in AB.VARIANTS.items()}
    assert len(set(libs.values()) | {base}) == len(libs) + 1, "every set of flags has a build of its own"
    monkeypatch.setenv("LMX_HOSTSIM", "1")
    monkeypatch.delenv("LMX_LIB_PATH", raising=False)
    table = AB.run_all(log=lambda *a: None, budget_s=600.0, small=True, base_lib=base, libs=libs)
    for group in AB.GROUPS:
        rows = table[group]
        assert set(rows) == {"base", "base_again"} | {n for n, v in AB.VARIANTS.items() if v[0] == group}
        for name, row in rows.items():
            assert "error" not in row and "skipped" not in row, (group, name, row)
            if not name.startswith("base"):
                assert row["results_equal_base"] is True, (group, name)
    keys = table["keys"]["base"]
    assert keys["split_state_0"]["pairs"] > 1000
    assert set(table["keys"]["keys_block_1024"]) >= {"split_state_0", "split_state_2"} and "split_state_1" not in table["keys"]["keys_block_1024"]
    for form in ("split_state_1", "split_state_2", "split_state_0_again"):  # the run-time forms of the mirror: same pairs, same carried state
        assert keys[form]["pairs_sha"] == keys["split_state_0"]["pairs_sha"] and keys[form]["state_sha"] == keys["split_state_0"]["state_sha"], form
    assert table["cull"]["base"]["all_test"]["visible"] > 0 and table["pose"]["base"]["pose_palette_us"] > 0


def test_ab_budget_and_missing_libraries_are_reported_not_raised(tmp_path):
    import ab_variants as AB

    table = AB.run_all(log=lambda *a: None, budget_s=0.0, small=True, base_lib=str(tmp_path / "nope.so"))
    for group in AB.GROUPS:
        assert all("skipped" in row for row in table[group].values()), table[group]
