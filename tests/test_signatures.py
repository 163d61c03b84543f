"""B1 (the Python face of the drop-in boundary) checked mechanically.  CPU only.

tests/golden/signatures.json is `inspect.signature` over the REFERENCE's three hot-path modules (written by
`tests/golden/make_golden.py signatures` in the build container) plus the state-dict key -> shape lists of the 5B transformer and
the default VAE built by the reference's constructors.  The mirrors in trajectorycrafter_amd/models are described by the same
function and compared:

  * every function / class / method / property the reference defines exists in the mirror, unless listed in NOT_MIRRORED with
    the reason;
  * every reference parameter exists under the same name, at the same position, of the same kind, with the same default; a
    mirror may ADD keyword parameters with defaults behind them and may give a default to a parameter the reference requires
    (supersets: every reference-style call still binds), never reorder, rename or change a default;
  * the mirrors' state dicts have exactly the reference's keys and shapes, and `.config` the reference's entries.
"""
import importlib
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import SIG_MODULES, describe_module        # noqa: E402

with open(os.path.join(HERE, "golden", "signatures.json")) as f:
    SIG = json.load(f)

# reference members the mirrors deliberately do not carry: (module, class or None, member) -> reason
NOT_MIRRORED = {
    ("models.crosstransformer3d", "CrossTransformer3DModel", "_set_gradient_checkpointing"): "training-only hook (gradient checkpointing); the path is inference",
    ("models.crosstransformer3d", "CrossTransformer3DModel", "_init_cross_inputs"): "constructor helper; the mirror builds the cross-attention layers in __init__",
    ("models.autoencoder_magvit", "CogVideoXEncoder3D", "_set_gradient_checkpointing"): "training-only hook",
    ("models.autoencoder_magvit", "CogVideoXDecoder3D", "_set_gradient_checkpointing"): "training-only hook",
    ("models.autoencoder_magvit", "AutoencoderKLCogVideoX", "_set_gradient_checkpointing"): "training-only hook",
}
# parameters whose default differs on purpose: (module, class, member, parameter) -> reason
DEFAULT_DIFFERS = {}


def _cmp_params(where, ref, got, problems):
    if ref is None:
        return
    if got is None:
        problems.append(f"{where}: no inspectable signature on the mirror")
        return
    gmap = {p[0]: (i, p) for i, p in enumerate(got)}
    for i, (name, kind, default) in enumerate(ref):
        if kind in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            continue
        if name not in gmap:
            problems.append(f"{where}: parameter `{name}` missing")
            continue
        j, (_, gkind, gdefault) = gmap[name]
        if kind == "POSITIONAL_OR_KEYWORD" and (j != i or gkind != kind):
            problems.append(f"{where}: parameter `{name}` is #{j} ({gkind}) on the mirror, #{i} ({kind}) in the reference")
        if kind == "KEYWORD_ONLY" and gkind not in ("KEYWORD_ONLY", "POSITIONAL_OR_KEYWORD"):
            problems.append(f"{where}: parameter `{name}` must be accepted by keyword")
        if default is None and gdefault is not None:
            continue                                      # required in the reference, optional on the mirror: every reference-style call still works
        if gdefault != default and (where, name) not in DEFAULT_DIFFERS:
            problems.append(f"{where}: default of `{name}` is {gdefault} on the mirror, {default} in the reference")
    rnames = {p[0] for p in ref}
    for name, kind, default in got:
        if name not in rnames and default is None and kind not in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            problems.append(f"{where}: extra parameter `{name}` without a default (a reference-style call would fail)")


@pytest.mark.parametrize("ref_name", sorted(SIG_MODULES))
def test_mirror_module_has_the_reference_call_surface(ref_name):
    ref = SIG["modules"][ref_name]
    got = describe_module(importlib.import_module(SIG_MODULES[ref_name]))
    problems = []
    for fn, params in ref["functions"].items():
        if fn not in got["functions"]:
            if (ref_name, None, fn) not in NOT_MIRRORED:
                problems.append(f"{fn}(): function missing")
            continue
        _cmp_params(f"{fn}()", params, got["functions"][fn], problems)
    for cls, desc in ref["classes"].items():
        if cls not in got["classes"]:
            if (ref_name, None, cls) not in NOT_MIRRORED:
                problems.append(f"class {cls} missing")
            continue
        gm = got["classes"][cls]["members"]
        for mname, m in desc["members"].items():
            if (ref_name, cls, mname) in NOT_MIRRORED:
                continue
            if mname not in gm:
                mirror_cls = getattr(importlib.import_module(SIG_MODULES[ref_name]), cls)
                if hasattr(mirror_cls, mname):           # provided by a base class of the mirror: still check the signature
                    import inspect
                    attr = inspect.getattr_static(mirror_cls, mname)
                    if m["kind"] == "property":
                        continue
                    fn = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
                    sig = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                           for p in inspect.signature(inspect.unwrap(fn)).parameters.values()]
                    _cmp_params(f"{cls}.{mname}()", m.get("params"), sig, problems)
                    continue
                problems.append(f"{cls}.{mname}: {m['kind']} missing")
                continue
            if m["kind"] == "property":
                if gm[mname]["kind"] != "property":
                    problems.append(f"{cls}.{mname}: a property in the reference, {gm[mname]['kind']} on the mirror")
                continue
            if gm[mname]["kind"] != m["kind"]:
                problems.append(f"{cls}.{mname}: {m['kind']} in the reference, {gm[mname]['kind']} on the mirror")
            _cmp_params(f"{cls}.{mname}()", m.get("params"), gm[mname].get("params"), problems)
    assert not problems, f"{len(problems)} deviations from the reference's {ref_name}:\n  " + "\n  ".join(problems)


def test_state_dict_keys_shapes_and_config_match_the_reference():
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    with torch.device("meta"):
        tr = CrossTransformer3DModel(**iw.TRANSFORMER_5B)
        vae = AutoencoderKLCogVideoX()
    for name, model in (("CrossTransformer3DModel(5B)", tr), ("AutoencoderKLCogVideoX()", vae)):
        ref = {k: tuple(v) for k, v in SIG["state_dict"][name].items()}
        got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert set(got) == set(ref), (name, sorted(set(ref) - set(got))[:10], sorted(set(got) - set(ref))[:10])
        assert all(got[k] == ref[k] for k in ref), [(k, got[k], ref[k]) for k in ref if got[k] != ref[k]][:10]
        rc = SIG["config"][name]
        gc = {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(model.config).items()}
        missing = [k for k in rc if k not in gc]
        differs = [(k, gc[k], rc[k]) for k in rc if k in gc and gc[k] != rc[k]]
        assert not missing and not differs, (name, missing, differs)
    # the random-init inventories (what bench / tests load) are those same keys
    assert set(iw.transformer_param_shapes(dict(tr.config))) == set(SIG["state_dict"]["CrossTransformer3DModel(5B)"])
    assert set(iw.vae_param_shapes(dict(vae.config))) == set(SIG["state_dict"]["AutoencoderKLCogVideoX()"])
