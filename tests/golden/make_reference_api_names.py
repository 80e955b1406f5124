"""The PUBLIC NAMES the reference's packages export, as data: the ``__all__`` of ``pysgmcmc.samplers``, ``pysgmcmc.diagnostics`` and
``pysgmcmc.models``, and the public top-level names of ``pysgmcmc.sampling`` (it has no ``__all__``). Read with ``ast`` (TensorFlow
is absent: the packages cannot be imported). Only the name lists travel (tests/golden/reference_api_names.json).

    python3 tests/golden/make_reference_api_names.py
"""
import ast
import json
import os

REF = "/root/reference/pysgmcmc"
HERE = os.path.dirname(os.path.abspath(__file__))


def dunder_all(path):
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "__all__" for t in node.targets):
            return [ast.literal_eval(e) for e in node.value.elts]
    return None


def public_toplevel(path):
    return [n.name for n in ast.parse(open(path).read()).body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and not n.name.startswith("_")]


names = {
    "samplers": dunder_all(os.path.join(REF, "samplers", "__init__.py")),
    "diagnostics": dunder_all(os.path.join(REF, "diagnostics", "__init__.py")),
    "models": dunder_all(os.path.join(REF, "models", "__init__.py")),
    "sampling": public_toplevel(os.path.join(REF, "sampling.py")),
    # modules on the sampler path without an __all__: their public top-level classes and functions
    "tensor_utils": public_toplevel(os.path.join(REF, "tensor_utils.py")),
    "stepsize_schedules": public_toplevel(os.path.join(REF, "stepsize_schedules.py")),
    "data_batches": public_toplevel(os.path.join(REF, "data_batches.py")),
    "source": {"samplers": "pysgmcmc/samplers/__init__.py:6-12", "diagnostics": "pysgmcmc/diagnostics/__init__.py:4-9",
               "models": "pysgmcmc/models/__init__.py:8-13", "sampling": "pysgmcmc/sampling.py:5 (class Sampler, no __all__)"},
}
json.dump(names, open(os.path.join(HERE, "reference_api_names.json"), "w"), indent=1)
print(names)
