"""Function-level rebinding of the reference's hot path.

chromosight's callers reach the path through module attributes -- `cid.pattern_detector(...)`,
`preproc.detrend(...)` (cli/chromosight.py:160-166, 243, 607, 791, 812; utils/contacts_map.py:15-17, 545, 610, 622) --
so a maintainer switches it to the MI355X by rebinding those attributes on the reference's own modules:

    import chromosight.utils.detection as cid
    import chromosight.utils.preprocessing as preproc
    import chromosight.utils.stats as cstats
    import chromosight_amd.integration
    chromosight_amd.integration.rebind(cid, preproc, cstats)

Swapping the modules wholesale would NOT work: `chromosight_amd.utils.preprocessing` deliberately has no
`get_detectable_bins` / `subsample_contacts` / `ztransform` / `sum_mat_bins` / `erase_missing` (out of the path, SURVEY.md
section 2), which `contacts_map.py:521, 584` call through the same `preproc` name.  KEEP lists what stays the
reference's; tests/test_integration_recipe.py checks both lists against the reference's call sites.
"""
from .utils import detection as _detection
from .utils import preprocessing as _preprocessing
from .utils import stats as _stats

# attribute names rebound on the reference's modules, by module
REBIND = {
    "detection": ["xcorr2", "normxcorr2", "pattern_detector", "pick_foci", "label_foci", "filter_foci", "validate_patterns",
                  "remove_neighbours", "pileup_patterns"],
    "preprocessing": ["detrend", "distance_law", "diag_trim", "set_mat_diag", "make_missing_mask", "frame_missing_mask",
                      "check_missing_mask", "zero_pad_sparse", "valid_to_missing", "factorise_kernel", "crop_kernel",
                      "resize_kernel"],
    "stats": ["fdr_correction", "corr_to_pval"],
}
# attributes of the same modules that chromosight's callers use and that stay the reference's own (outside the hot path)
KEEP = {
    "detection": [],
    "preprocessing": ["get_detectable_bins", "subsample_contacts", "ztransform", "sum_mat_bins", "erase_missing"],
    "stats": [],
}
_OURS = {"detection": _detection, "preprocessing": _preprocessing, "stats": _stats}


def rebind(detection=None, preprocessing=None, stats=None):
    """Point the hot-path functions of the given reference modules at this package's.  Returns the
    {module: {name: previous function}} of what was replaced, so that `restore` can undo it."""
    previous = {}
    for key, module in (("detection", detection), ("preprocessing", preprocessing), ("stats", stats)):
        if module is None:
            continue
        previous[key] = {}
        for name in REBIND[key]:
            previous[key][name] = getattr(module, name, None)
            setattr(module, name, getattr(_OURS[key], name))
    return previous


def restore(previous, detection=None, preprocessing=None, stats=None):
    for key, module in (("detection", detection), ("preprocessing", preprocessing), ("stats", stats)):
        if module is None or key not in previous:
            continue
        for name, fn in previous[key].items():
            if fn is None:
                delattr(module, name)
            else:
                setattr(module, name, fn)
