"""Record / replay of the reference's call sequence on the skeleton objects (test infrastructure).

``tests/golden/record_api.py`` (build container only: it imports the reference) runs the REFERENCE's own callers —
``SkeletonModel``'s methods, ``TrainRig.train_step`` and everything under it, ``render_rig.render_set`` /
``generate_random_motion``, ``GUI.test_step`` — against the reference's ``SkeletonWarp`` wrapped in recording proxies, and
dumps every attribute read / attribute write / call they make (who made it: reference file:line; argument and result
descriptions with the values of small tensors) to ``tests/golden/skeleton_api_calls.json``.  This module holds what both
sides share and imports nothing of the reference: the value descriptions, the seeded parameter values, the proxies, and the
replay of a recorded list on other objects (the HIP classes) with a comparison of every result.
"""
from __future__ import annotations

import base64
import math
import types
import zlib

import numpy as np

import torch
import torch.nn as nn

MAX_STORED = 4096  # tensors up to this many elements travel by value
SHAPES_ONLY = ("state_dict", "load_state_dict", "trainable_parameters", "parameters")  # calls described without values


# --------------------------------------------------------------------------- seeded state both sides agree on
def seeded_values(name: str, shape, seed: int) -> torch.Tensor:
    """Deterministic values for the parameter called ``name`` — the same on the reference's module and on the mirror, so
    that recorded results can be compared by value without shipping a 2 MB state dict."""
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    shape = tuple(shape)
    if name == "_node_radius":
        return math.log(0.15) + 0.3 * torch.randn(shape, generator=g)
    if name == "control_nodes":
        return 0.3 * torch.randn(shape, generator=g)
    if name.endswith("gaussian_warp.weight"):
        return 1e-2 * torch.randn(shape, generator=g)
    if len(shape) >= 2:
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[-1])
    return 0.01 * torch.randn(shape, generator=g)


def seed_module(module: nn.Module, seed: int, hyper_from: int = 3, keep=()):
    """Overwrite every parameter the two implementations share (everything but the stage-1 ``network.*`` leftovers); the
    joints (``nodes[:, :3]``) stay, the hyper coordinates behind them are seeded."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.startswith("network.") or name in keep:
                continue
            if name == "nodes":
                v = seeded_values(name, (p.shape[0], p.shape[1] - hyper_from), seed)
                p.data[:, hyper_from:] = v.to(p.device)
                continue
            p.data.copy_(seeded_values(name, p.shape, seed).to(p.device))


# --------------------------------------------------------------------------- descriptions
def _is_namespace(v):
    return isinstance(v, (types.SimpleNamespace,)) or type(v).__name__ in ("Namespace", "GroupParams")


def describe(v, depth=0, store=True):
    if v is None:
        return {"t": "none"}
    if isinstance(v, torch.Tensor):
        d = {"t": "tensor", "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
             "param": isinstance(v, nn.Parameter), "grad": bool(v.requires_grad)}
        if store and v.numel() <= MAX_STORED:
            d["b64"] = base64.b64encode(v.detach().cpu().contiguous().numpy().tobytes()).decode()  # exact, little-endian
        return d
    if isinstance(v, bool):
        return {"t": "bool", "v": v}
    if isinstance(v, int):
        return {"t": "int", "v": v}
    if isinstance(v, float):
        return {"t": "float", "v": v}
    if isinstance(v, str):
        return {"t": "str", "v": v}
    if isinstance(v, dict):
        return {"t": "dict", "items": {str(k): describe(v[k], depth + 1, store) for k in v.keys()}}  # v[k]: lazy dicts fill on access
    if isinstance(v, (list, tuple)):
        if depth < 3 and len(v) <= 64:
            return {"t": "list", "items": [describe(x, depth + 1, store) for x in v]}
        return {"t": "list", "len": len(v)}
    if isinstance(v, torch.optim.Optimizer):
        return {"t": "optimizer", "groups": [{"name": g.get("name"), "lr": float(g["lr"]), "n_params": len(g["params"]),
                                             "numel": sum(int(p.numel()) for p in g["params"])} for g in v.param_groups]}
    if isinstance(v, nn.Module):
        return {"t": "module"}
    if _is_namespace(v):
        fields = {k: x for k, x in vars(v).items() if isinstance(x, (bool, int, float, str)) and not k.startswith("_")}
        return {"t": "namespace", "fields": fields}
    if callable(v):
        return {"t": "callable"}
    return {"t": "object"}


def _stored(d):
    """The stored values of a tensor description (a CPU tensor of its dtype and shape)."""
    a = np.frombuffer(base64.b64decode(d["b64"]), dtype=np.dtype(d["dtype"] if d["dtype"] != "bool" else "bool_")).reshape(d["shape"])
    return torch.from_numpy(a.copy())


def rebuild(d, device):
    """A value from its description (arguments of a replayed call)."""
    t = d["t"]
    if t == "none":
        return None
    if t == "tensor":
        if "b64" not in d:
            raise ValueError("argument tensor without stored values")
        x = _stored(d).to(device)
        if d["grad"] and x.is_floating_point():
            x.requires_grad_(True)
        return x
    if t in ("bool", "int", "float", "str"):
        return d["v"]
    if t == "dict":
        return {k: rebuild(x, device) for k, x in d["items"].items()}
    if t == "list":
        return [rebuild(x, device) for x in d["items"]]
    if t == "namespace":
        return types.SimpleNamespace(**d["fields"])
    raise ValueError("cannot rebuild a %s argument" % t)


def _foreign(key):
    """State-dict entries of the stage-1 network a reference ``SkeletonWarp`` drags along (utils/time_utils.py:797-804; never
    evaluated on the skeleton path) — the mirror keeps one placeholder entry instead."""
    return str(key).startswith("network.")


def mismatches(ref, mine, what, rtol=1e-4, atol=1e-5):
    """Differences between a recorded description and a live value (empty list = same)."""
    out = []
    t = ref["t"]
    if t == "none":
        if mine is not None:
            out.append("%s: expected None, got %s" % (what, type(mine).__name__))
    elif t == "tensor":
        if not isinstance(mine, torch.Tensor):
            return ["%s: expected a tensor, got %s" % (what, type(mine).__name__)]
        if list(mine.shape) != ref["shape"]:
            out.append("%s: shape %s, reference %s" % (what, list(mine.shape), ref["shape"]))
        elif str(mine.dtype).replace("torch.", "") != ref["dtype"]:
            out.append("%s: dtype %s, reference %s" % (what, mine.dtype, ref["dtype"]))
        elif "b64" in ref:
            r = _stored(ref)
            m = mine.detach().cpu()
            if mine.is_floating_point():
                bad = (m - r).abs() > atol + rtol * r.abs().max().clamp_min(1e-30)
                if bool(bad.any()) or not bool(torch.isfinite(m).all()):
                    out.append("%s: %d of %d values differ (max |err| %.3g, reference max %.3g)"
                               % (what, int(bad.sum()), m.numel(), float((m - r).abs().max()), float(r.abs().max())))
            elif not torch.equal(m, r):
                out.append("%s: integer values differ" % what)
        if ref["param"] and not isinstance(mine, nn.Parameter):
            out.append("%s: the reference returns an nn.Parameter" % what)
    elif t in ("bool", "int", "str"):
        if type(mine).__name__ != t or mine != ref["v"]:
            out.append("%s: %r, reference %r" % (what, mine, ref["v"]))
    elif t == "float":
        if not isinstance(mine, (int, float)) or abs(float(mine) - ref["v"]) > atol + rtol * abs(ref["v"]):
            out.append("%s: %r, reference %r" % (what, mine, ref["v"]))
    elif t == "dict":
        if not isinstance(mine, dict):
            return ["%s: expected a dict, got %s" % (what, type(mine).__name__)]
        if {k for k in map(str, mine.keys()) if not _foreign(k)} != {k for k in ref["items"].keys() if not _foreign(k)}:
            out.append("%s: keys %s, reference %s" % (what, sorted(map(str, mine.keys())), sorted(ref["items"].keys())))
        for k, x in ref["items"].items():
            if k in mine:
                out += mismatches(x, mine[k], "%s[%r]" % (what, k), rtol, atol)
    elif t == "list":
        if not isinstance(mine, (list, tuple)):
            return ["%s: expected a list, got %s" % (what, type(mine).__name__)]
        n = len(ref["items"]) if "items" in ref else ref["len"]
        if len(mine) != n:
            out.append("%s: %d entries, reference %d" % (what, len(mine), n))
        elif "items" in ref:
            for i, x in enumerate(ref["items"]):
                out += mismatches(x, mine[i], "%s[%d]" % (what, i), rtol, atol)
    elif t == "optimizer":
        if not isinstance(mine, torch.optim.Optimizer):
            return ["%s: expected an optimizer, got %s" % (what, type(mine).__name__)]
        got = describe(mine)["groups"]
        if [(g["name"], g["n_params"], g["numel"]) for g in got] != [(g["name"], g["n_params"], g["numel"]) for g in ref["groups"]]:
            out.append("%s: groups %s, reference %s" % (what, [(g["name"], g["numel"]) for g in got],
                                                        [(g["name"], g["numel"]) for g in ref["groups"]]))
        else:
            for a, b in zip(got, ref["groups"]):
                if abs(a["lr"] - b["lr"]) > 1e-12 + 1e-6 * abs(b["lr"]):
                    out.append("%s: lr of group %s is %g, reference %g" % (what, a["name"], a["lr"], b["lr"]))
    elif t == "module":
        if not isinstance(mine, nn.Module):
            out.append("%s: expected an nn.Module, got %s" % (what, type(mine).__name__))
    elif t == "callable":
        if not callable(mine):
            out.append("%s: expected a callable" % what)
    elif t in ("object", "namespace"):
        if mine is None:
            out.append("%s: expected an object, got None" % what)
    return out


# --------------------------------------------------------------------------- recording proxies
class Recorder:
    """Wraps an object: every attribute read, attribute write and call made THROUGH the wrapper is appended to ``log``
    with the caller's position.  ``children`` maps attribute names to the path their value is wrapped under (so that
    ``skeleton.deform`` and ``deform.as_gaussians`` record too)."""

    def __init__(self, target, path, log, locate, children=None):
        object.__setattr__(self, "_r", (target, path, log, locate, children or {}))

    def _wrap(self, name, value):
        target, path, log, locate, children = object.__getattribute__(self, "_r")
        if name in children and value is not None and not isinstance(value, Recorder):
            sub_path, sub_children = children[name]
            return Recorder(value, sub_path, log, locate, sub_children), sub_path
        return value, (object.__getattribute__(value, "_r")[1] if isinstance(value, Recorder) else None)

    def __getattr__(self, name):
        target, path, log, locate, children = object.__getattribute__(self, "_r")
        value = getattr(target, name)
        if isinstance(value, types.MethodType) or (callable(value) and not isinstance(value, (nn.Module, torch.Tensor, Recorder))
                                                   and not isinstance(value, type)):
            def call(*a, **k):
                st = name not in SHAPES_ONLY
                ev = {"who": locate(), "path": path, "op": "call", "name": name, "args": [describe(unwrap(x), store=st) for x in a],
                      "kwargs": {kk: describe(unwrap(x), store=st) for kk, x in k.items()}}
                log.append(ev)
                res = value(*[unwrap(x) for x in a], **{kk: unwrap(x) for kk, x in k.items()})
                ev["result"] = describe(res, store=st)
                return res
            return call
        wrapped, sub = self._wrap(name, value)
        ev = {"who": locate(), "path": path, "op": "get", "name": name, "result": describe(value if sub is None else unwrap(value))}
        if sub is not None:
            ev["binds"] = sub
        log.append(ev)
        return wrapped

    def __setattr__(self, name, value):
        target, path, log, locate, children = object.__getattribute__(self, "_r")
        log.append({"who": locate(), "path": path, "op": "set", "name": name, "value": describe(unwrap(value))})
        setattr(target, name, unwrap(value))

    def __call__(self, *a, **k):
        target, path, log, locate, children = object.__getattribute__(self, "_r")
        ev = {"who": locate(), "path": path, "op": "call", "name": "__call__", "args": [describe(unwrap(x)) for x in a],
              "kwargs": {kk: describe(unwrap(x)) for kk, x in k.items()}}
        log.append(ev)
        res = target(*[unwrap(x) for x in a], **{kk: unwrap(x) for kk, x in k.items()})
        ev["result"] = describe(res)
        return res


def unwrap(v):
    return object.__getattribute__(v, "_r")[0] if isinstance(v, Recorder) else v


# --------------------------------------------------------------------------- replay
def replay(events, objects, device, skip=lambda ev: False, rtol=1e-4, atol=1e-5):
    """Apply ``events`` to ``objects`` (path -> object; paths bound by recorded reads are added on the way).  Returns
    (number applied, list of differences)."""
    problems, applied = [], 0
    for i, ev in enumerate(events):
        if skip(ev):
            continue
        path, op, name = ev["path"], ev["op"], ev["name"]
        tag = "#%d %s.%s [%s] (%s)" % (i, path, name, op, ev["who"])
        if path not in objects:
            problems.append("%s: nothing bound to path %r yet" % (tag, path))
            continue
        obj = objects[path]
        applied += 1
        try:
            if op == "get":
                val = getattr(obj, name)
                if "binds" in ev:
                    objects[ev["binds"]] = val
                else:
                    problems += mismatches(ev["result"], val, tag, rtol, atol)
            elif op == "set":
                setattr(obj, name, rebuild(ev["value"], device))
            elif op == "set_data":
                getattr(obj, name).data = rebuild(ev["value"], device)
            elif op == "call" and name == "load_state_dict":
                # a checkpoint with the REFERENCE's keys and shapes (its stage-1 ``network.*`` leftovers, ``gs_*`` ...): the
                # entries this object knows keep their current values, the others are filled
                mine = obj.state_dict()
                sd = {k: (mine[k].detach().clone() if k in mine else seeded_values(k, d["shape"], 0).to(device))
                      for k, d in ev["args"][0]["items"].items()}
                obj.load_state_dict(sd)
            elif op == "call":
                args = [rebuild(a, device) for a in ev["args"]]
                kwargs = {k: rebuild(a, device) for k, a in ev["kwargs"].items()}
                fn = obj if name == "__call__" else getattr(obj, name)
                res = fn(*args, **kwargs)
                if "result" in ev:
                    problems += mismatches(ev["result"], res, tag, rtol, atol)
            else:
                problems.append("%s: unknown op" % tag)
        except Exception as e:  # an AttributeError here is exactly the failure the recording exists to catch
            problems.append("%s: raised %s: %s" % (tag, type(e).__name__, e))
    return applied, problems
