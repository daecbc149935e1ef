"""User environments: a C++ MODEL HEADER (include/ctk_user_env.h) compiled at configure time into a library of its own, with every
optimizer kernel instantiated for it — no file under csrc/ is edited.

The reference selects plant model and concrete cost at run time — `PredictorWrapper()` + `predictor_specification`
(Controllers/controller_mpc.py:43,67-73), `Control_Toolkit_ASF.Cost_Functions.<environment>.<name>` imported by name
(Cost_Functions/cost_function_wrapper.py:59-66) — and compiles at configure time itself where it needs native code
(Controllers/controller_C.py:140-248).  Here:

    from control_toolkit_amd.build_env import register_environment
    name = register_environment("path/to/my_env.h")     # hipcc --offload-arch=gfx950, ~1-2 minutes the first time; cached by content
    CtkEngine("mppi", "ODE", environment=name, ...)      # or controller_mpc(..., environment_name=name)

The library lands in control_toolkit_amd/_env_builds/<Name>_<digest>/ (in-tree: it travels with the tree like libctk_hip.so; the digest covers
the header AND the kernel sources, so a stale build is never loaded).  It is the whole engine with a fourth environment id
(CTK_ENV_USER): the built environments work in it too."""
import hashlib
import os
import re
import subprocess

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_OUT = os.path.join(_HERE, "_env_builds")


def _digest(header_text: bytes) -> str:
    h = hashlib.sha256(header_text)
    files = sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".h", ".inc")) or f == "Makefile")
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(_CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_HERE), "include", "ctk_hip.h"), "rb").read())
    return h.hexdigest()[:12]


def environment_name_of(header_path: str) -> str:
    """the NAME the model declares (`static constexpr const char* NAME = "...";`)"""
    text = open(header_path, "r").read()
    m = re.search(r'\bNAME\s*=\s*"([A-Za-z][A-Za-z0-9_]*)"', text)
    if not m:
        raise ValueError(f"{header_path}: no `static constexpr const char* NAME = \"...\";` in struct CtkUserEnv (include/ctk_user_env.h)")
    return m.group(1)


def build_environment(header_path: str, jobs: int = 8, verbose: bool = False):
    """compile (or find in the cache) the library for the model header; returns (name, library path)"""
    header_path = os.path.abspath(header_path)
    name = environment_name_of(header_path)
    if name in _capi.ENVIRONMENTS:
        raise ValueError(f"{header_path}: the name {name!r} is a built environment's")
    out_dir = os.path.join(_OUT, f"{name}_{_digest(open(header_path, 'rb').read())}")
    lib = os.path.join(out_dir, f"libctk_hip_{name}.so")
    if not os.path.exists(lib):
        import fcntl
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, ".lock"), "w") as lock:          # two processes asking for the same model: one builds, the other waits
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(lib):
                if any(ch in out_dir for ch in " '\""):
                    raise ValueError(f"{out_dir}: the build path is handed to make / the shell unquoted; no spaces or quotes in it, please")
                # the build reads a COPY of the header next to the objects: the library stays valid when the original moves
                copy = os.path.join(out_dir, "user_env.h")
                with open(copy, "wb") as f:
                    f.write(open(header_path, "rb").read())
                cmd = ["make", "-C", _CSRC, f"-j{max(1, min(jobs, os.cpu_count() or 1))}", f"BUILD={os.path.join(out_dir, 'obj')}", f"LIB={lib}",
                       f"EXTRA=-DCTK_USER_ENV_HEADER='\"{copy}\"'", f"USER_ENV_DEP={copy}"]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0 or not os.path.exists(lib):
                    tail = "\n".join((r.stdout + "\n" + r.stderr).splitlines()[-40:])
                    raise RuntimeError(f"building the environment {name!r} from {header_path} failed (hipcc):\n{tail}")
                if verbose:
                    print(r.stdout[-2000:])
                # the objects are not needed again (another digest = another directory), and builds of this model against OLDER kernel
                # sources can never be loaded again: drop both (best effort — another process may hold one of them open)
                import shutil
                shutil.rmtree(os.path.join(out_dir, "obj"), ignore_errors=True)
                for other in os.listdir(_OUT):
                    if other.startswith(name + "_") and os.path.join(_OUT, other) != out_dir:
                        shutil.rmtree(os.path.join(_OUT, other), ignore_errors=True)
    return name, lib


def register_environment(header_path: str, jobs: int = 8) -> str:
    """build_environment + make `environment=<name>` / `environment_name: <name>` resolve to it in this process; returns the name"""
    name, lib = build_environment(header_path, jobs=jobs)
    _capi.USER_ENVIRONMENTS[name] = lib
    S, C, _ = _capi.environment_info(name)            # loads the library: fails here, loudly, if it does not carry the environment
    # the host-side wrappers: a user model declares ONE flat parameter list (dynamics constants and cost weights alike); all of it is
    # offered through the cost wrapper's dictionary (set per step by name like the built environments' weights), the predictor wrapper adds none
    from . import Predictors, Cost_Functions
    Predictors.ENVIRONMENT_DIMS[name] = (S, C)
    Predictors.DEFAULT_DYNAMICS_BY_ENV[name] = {}
    Cost_Functions.DEFAULT_COST_BY_ENV[name] = _capi.environment_defaults(name)
    Cost_Functions.DEFAULT_ATTRIBUTES_BY_ENV[name] = {}
    return name
