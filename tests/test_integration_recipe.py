"""the function-level rebinding recipe (chromosight_amd.integration) against the attributes the
reference's callers actually use (tests/golden/reference_call_sites.json, parsed from cli/chromosight.py and
utils/contacts_map.py by tests/golden/make_call_sites.py).  CPU only."""
import importlib
import inspect
import json
import pathlib
import types

import pytest

from chromosight_amd import integration

HERE = pathlib.Path(__file__).resolve().parent
SITES = json.loads((HERE / "golden" / "reference_call_sites.json").read_text())


def test_every_attribute_the_callers_use_is_rebound_or_kept():
    for module, attrs in SITES.items():
        for name, where in attrs.items():
            assert name in integration.REBIND[module] or name in integration.KEEP[module], (module, name, where)
    # the call sites the boundary is about
    assert "pattern_detector" in SITES["detection"] and "remove_neighbours" in SITES["detection"]
    assert {"detrend", "diag_trim", "valid_to_missing"} <= set(SITES["preprocessing"])
    # what a module swap would have broken: used by the callers, deliberately not in this package
    assert {"get_detectable_bins", "subsample_contacts"} <= set(SITES["preprocessing"])
    assert {"get_detectable_bins", "subsample_contacts"} <= set(integration.KEEP["preprocessing"])


def test_rebound_names_exist_here_and_kept_names_do_not_shadow():
    for module, names in integration.REBIND.items():
        ours = importlib.import_module(f"chromosight_amd.utils.{module}")
        for name in names:
            assert callable(getattr(ours, name, None)), (module, name)
        assert not set(names) & set(integration.KEEP[module])


def test_rebind_and_restore_on_stand_in_modules():
    mods = {k: types.SimpleNamespace(**{n: (lambda *a, _n=n: _n) for n in integration.REBIND[k] + integration.KEEP[k]})
            for k in integration.REBIND}
    kept = {k: {n: getattr(mods[k], n) for n in integration.KEEP[k]} for k in mods}
    prev = integration.rebind(**mods)
    import chromosight_amd.utils.detection as cud
    import chromosight_amd.utils.preprocessing as cup
    assert mods["detection"].pattern_detector is cud.pattern_detector
    assert mods["preprocessing"].detrend is cup.detrend
    for k in mods:
        for n in integration.KEEP[k]:
            assert getattr(mods[k], n) is kept[k][n]            # untouched
    integration.restore(prev, **mods)
    assert mods["detection"].pattern_detector() == "pattern_detector"


def test_signatures_match_the_reference_where_it_is_present():
    """In the authoring container the reference imports: same parameter names, order and defaults for every rebound
    function, and the fixture is current."""
    ref_root = pathlib.Path("/root/reference")
    if not (ref_root / "chromosight").exists():
        pytest.skip("the reference does not travel to this box")
    import sys
    sys.path.insert(0, str(ref_root))
    try:
        for module, names in integration.REBIND.items():
            ref = importlib.import_module(f"chromosight.utils.{module}")
            ours = importlib.import_module(f"chromosight_amd.utils.{module}")
            for name in names:
                a = inspect.signature(getattr(ref, name)).parameters
                b = inspect.signature(getattr(ours, name)).parameters
                assert list(a) == list(b), (module, name, list(a), list(b))
                for p in a:
                    da, db = a[p].default, b[p].default
                    assert (da is inspect._empty) == (db is inspect._empty), (module, name, p)
                    if da is not inspect._empty and not callable(da):
                        assert da == db, (module, name, p, da, db)
    finally:
        sys.path.remove(str(ref_root))
    import subprocess
    out = subprocess.run([sys.executable, str(HERE / "golden" / "make_call_sites.py")], capture_output=True, text=True, check=True)
    assert json.loads(out.stdout) == SITES
