"""The drop-in boundary against the reference itself (build container only; skipped where /root/reference is absent, e.g.
on the GPU box -- nothing of the reference travels).

SURVEY 8(b): the reference has no FFI on this path; its boundary is the Python import surface of
ref:src/train/cli/train_v33_ddp.py:35-50.  The reference files are read as TEXT and compared through `ast` -- no
reference code is imported or executed in the test process: every `from src.... import name` of the
reference CLI must resolve to an attribute of this repo's module of the same dotted name, the eight trainer functions the
CLI inlines (ref:train_v33_ddp.py:105-448; their home here is src.train.core.ddp_trainer, as north_star names it) must
have the reference's parameter lists, this repo's CLI must accept the reference's flags with the same type / default /
action, and the V33*Config dataclasses must have the reference's fields and defaults (ref:src/train/config/v33.py:21-132)."""
import ast
import dataclasses
import importlib
import inspect
import os

import pytest

REF = "/root/reference"
CLI = os.path.join(REF, "src/train/cli/train_v33_ddp.py")
CFG = os.path.join(REF, "src/train/config/v33.py")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="reference checkout not present (build container only)")


def _tree(path):
    with open(path, encoding="utf-8") as fh:
        return ast.parse(fh.read(), filename=path)


def _sig_from_ast(fn: ast.FunctionDef):
    """[(name, kind, default-as-source)] of a function definition, as inspect.signature would list it."""
    a = fn.args
    out = []
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for i, (arg, d) in enumerate(zip(pos, defaults)):
        kind = "POSITIONAL_ONLY" if i < len(a.posonlyargs) else "POSITIONAL_OR_KEYWORD"
        out.append((arg.arg, kind, "<required>" if d is None else ast.unparse(d)))
    if a.vararg:
        out.append((a.vararg.arg, "VAR_POSITIONAL", "<required>"))
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append((arg.arg, "KEYWORD_ONLY", "<required>" if d is None else ast.unparse(d)))
    if a.kwarg:
        out.append((a.kwarg.arg, "VAR_KEYWORD", "<required>"))
    return out


def _sig_live(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        d = "<required>" if p.default is inspect.Parameter.empty else repr(p.default)
        out.append((p.name, p.kind.name, d))
    return out


def _functions(tree):
    return {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}


def _flags(tree):
    """{'--flag': {'type': ..., 'default': ..., 'action': ...}} of every parser.add_argument(...) in parse_args()."""
    fn = _functions(tree)["parse_args"]
    out = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            names = [ast.literal_eval(a) for a in node.args]
            kw = {k.arg: k.value for k in node.keywords}
            rec = {"type": ast.unparse(kw["type"]) if "type" in kw else None,
                   "default": ast.literal_eval(kw["default"]) if "default" in kw else None,
                   "action": ast.literal_eval(kw["action"]) if "action" in kw else None}
            out[names[-1]] = rec
    return out


def test_reference_cli_imports_resolve_here_and_signatures_match():
    import src                                               # this repo's mirror package (tests/conftest.py put it on sys.path)
    here = os.path.dirname(os.path.dirname(os.path.abspath(src.__file__)))
    assert "opensearch-neural-pre-train_amd" in here and not here.startswith(REF)
    ref = _tree(CLI)
    # every `from src.x.y import a, b` of the reference CLI (module level, and inside its try/except for optional ones)
    wanted, optional = [], []
    for node in ast.walk(ref):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "src" and node.level == 0:
            in_try = any(isinstance(p, ast.Try) and node in ast.walk(p) for p in ref.body)
            (optional if in_try else wanted).extend((node.module, a.name) for a in node.names)
    assert ("src.model.splade_modern", "SPLADEModernBERT") in wanted and ("src.model.losses", "SPLADELossV33") in wanted
    for mod, name in wanted:
        m = importlib.import_module(mod)
        assert not os.path.abspath(m.__file__).startswith(REF), mod
        assert hasattr(m, name), f"{mod}.{name} (imported by the reference CLI) is missing here"
    for mod, name in optional:                               # ref:train_v33_ddp.py:46-49: may stay absent
        try:
            m = importlib.import_module(mod)
        except ImportError:
            continue
        assert hasattr(m, name), f"{mod} exists here but lacks {name}"
    # the trainer functions: same names, same parameter lists (names, kinds, defaults)
    from src.train.core import ddp_trainer as T
    ref_fns = _functions(ref)
    for n in ("setup_distributed", "cleanup_distributed", "is_main_process", "create_dataloader_ddp", "save_checkpoint",
              "load_checkpoint", "find_latest_checkpoint", "train_epoch"):
        assert _sig_live(getattr(T, n)) == _sig_from_ast(ref_fns[n]), (n, _sig_live(getattr(T, n)), _sig_from_ast(ref_fns[n]))
    # and this repo's own CLI declares the reference's flags (same type, default, action)
    from src.train.cli import train_v33_ddp as our_cli
    ours = _flags(_tree(inspect.getsourcefile(our_cli)))
    theirs = _flags(ref)
    assert len(theirs) == 12
    missing = {k: v for k, v in theirs.items() if ours.get(k) != v}
    assert not missing, missing


def _ref_dataclasses(tree):
    """{class name: [(field, default)]}: literals evaluated; field(default_factory=lambda: <literal>) -> the literal;
    field(default_factory=ClassName) -> ('<instance>', ClassName); no default -> '<required>'."""
    out = {}
    for node in tree.body:
        if not isinstance(node, ast.ClassDef):
            continue
        fields = []
        for st in node.body:
            if not isinstance(st, ast.AnnAssign) or not isinstance(st.target, ast.Name):
                continue
            v = st.value
            if v is None:
                d = "<required>"
            elif isinstance(v, ast.Call) and ast.unparse(v.func) in ("field", "dataclasses.field"):
                kw = {k.arg: k.value for k in v.keywords}
                if "default" in kw:
                    d = ast.literal_eval(kw["default"])
                else:
                    fac = kw["default_factory"]
                    if isinstance(fac, ast.Lambda):
                        d = ast.literal_eval(fac.body)
                    elif isinstance(fac, ast.Name) and fac.id in ("list", "dict"):
                        d = [] if fac.id == "list" else {}
                    else:
                        d = ("<instance>", ast.unparse(fac))
            else:
                d = ast.literal_eval(v)
            fields.append((st.target.id, d))
        out[node.name] = fields
    return out


def test_config_dataclasses_equal_the_references():
    ref = _ref_dataclasses(_tree(CFG))
    from src.train.config import v33 as ours
    for n in ("V33Config", "V33DataConfig", "V33LossConfig", "V33ModelConfig", "V33TrainingConfig"):
        fa = ref[n]
        fb = [(f.name, _default(f)) for f in dataclasses.fields(getattr(ours, n))]
        assert [x[0] for x in fa] == [x[0] for x in fb], n
        for (name, da), (_, db) in zip(fa, fb):
            if isinstance(da, tuple) and da and da[0] == "<instance>":
                assert dataclasses.is_dataclass(db) and type(db).__name__ == da[1], (n, name, da, db)
                continue
            assert da == db and type(da) is type(db), (n, name, da, db)


def _default(f):
    if f.default is not dataclasses.MISSING:
        return f.default
    if f.default_factory is not dataclasses.MISSING:
        return f.default_factory()
    return "<required>"
