"""The drop-in boundary against the reference itself (build container only; skipped where /root/reference is absent, e.g.
on the GPU box -- nothing of the reference travels).

SURVEY 8(b): the reference has no FFI on this path; its boundary is the Python import surface of
ref:src/train/cli/train_v33_ddp.py:35-50.  This test loads that file BY PATH, unmodified, with this repo's package
directory first on sys.path: every `from src....` import of the reference CLI must resolve to this repo's modules, the
eight trainer functions the CLI inlines (ref:train_v33_ddp.py:105-448; their home here is src.train.core.ddp_trainer, as
north_star names it) must have the reference's parameter lists, and the V33*Config dataclasses must have the reference's
fields and defaults (ref:src/train/config/v33.py:21-132)."""
import dataclasses
import importlib.util
import inspect
import os
import sys

import pytest

REF = "/root/reference"
CLI = os.path.join(REF, "src/train/cli/train_v33_ddp.py")
CFG = os.path.join(REF, "src/train/config/v33.py")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="reference checkout not present (build container only)")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _params(fn):
    return [(p.name, p.kind, p.default if p.default is not inspect.Parameter.empty else "<required>")
            for p in inspect.signature(fn).parameters.values()]


def test_reference_cli_imports_against_this_package_and_signatures_match():
    import src                                               # this repo's mirror package (tests/conftest.py put it on sys.path)
    here = os.path.dirname(os.path.dirname(os.path.abspath(src.__file__)))
    assert "opensearch-neural-pre-train_amd" in here and not here.startswith(REF)
    ref_cli = _load(CLI, "_ref_train_v33_ddp")               # executes the reference's import block against OUR src.*
    # what the reference CLI bound at import time are this repo's objects
    from src.model.losses import SPLADELossV33
    from src.model.splade_modern import SPLADEModernBERT
    from src.train.core import ddp_trainer as T
    from src.train.data.dataloader import TripletCollator
    assert ref_cli.SPLADEModernBERT is SPLADEModernBERT and ref_cli.SPLADELossV33 is SPLADELossV33
    assert ref_cli.TripletCollator is TripletCollator
    for n in ("V33Config", "V33DataConfig", "V33LossConfig", "V33ModelConfig", "V33TrainingConfig", "load_training_data",
              "create_tokenizer", "TensorBoardLogger", "setup_logging"):
        assert getattr(ref_cli, n).__module__.startswith("src."), n
        assert not inspect.getsourcefile(getattr(ref_cli, n)).startswith(REF), n
    # the trainer functions: same names, same parameter lists (names, kinds, defaults)
    for n in ("setup_distributed", "cleanup_distributed", "is_main_process", "create_dataloader_ddp", "save_checkpoint",
              "load_checkpoint", "find_latest_checkpoint", "train_epoch"):
        assert _params(getattr(T, n)) == _params(getattr(ref_cli, n)), (n, _params(getattr(T, n)), _params(getattr(ref_cli, n)))
    # and this repo's own CLI parses the reference's flags
    from src.train.cli import train_v33_ddp as our_cli
    ref_flags = {a.option_strings[-1]: (a.default, a.type) for a in _parser_actions(ref_cli)}
    our_flags = {a.option_strings[-1]: (a.default, a.type) for a in _parser_actions(our_cli)}
    missing = {k: v for k, v in ref_flags.items() if our_flags.get(k) != v}
    assert not missing, missing


def _parser_actions(mod):
    import argparse
    seen = []
    orig = argparse.ArgumentParser.parse_args

    def grab(self, *a, **k):
        seen.extend(x for x in self._actions if x.option_strings and x.dest != "help")
        return argparse.Namespace()
    argparse.ArgumentParser.parse_args = grab
    try:
        mod.parse_args()
    finally:
        argparse.ArgumentParser.parse_args = orig
    return seen


def test_config_dataclasses_equal_the_references():
    ref_cfg = _load(CFG, "_ref_config_v33")
    from src.train.config import v33 as ours
    for n in ("V33Config", "V33DataConfig", "V33LossConfig", "V33ModelConfig", "V33TrainingConfig"):
        a, b = getattr(ref_cfg, n), getattr(ours, n)
        fa = [(f.name, _default(f)) for f in dataclasses.fields(a)]
        fb = [(f.name, _default(f)) for f in dataclasses.fields(b)]
        assert [x[0] for x in fa] == [x[0] for x in fb], n
        for (name, da), (_, db) in zip(fa, fb):
            if dataclasses.is_dataclass(da) and dataclasses.is_dataclass(db):
                da, db = dataclasses.asdict(da), dataclasses.asdict(db)
            assert da == db, (n, name, da, db)


def _default(f):
    if f.default is not dataclasses.MISSING:
        return f.default
    if f.default_factory is not dataclasses.MISSING:
        return f.default_factory()
    return "<required>"
