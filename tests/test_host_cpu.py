"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/ctk_hip.h
declares, the ctypes struct matches the C struct, and the host-side mirror of the reference's
plugin interface behaves like the reference (names, argument meaning, error behaviour).  No
compute call is made: without a GPU the engine must fail loudly, never fall back."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ctk_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctk_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from control_toolkit_amd._capi import load_library, SYMBOLS, library_path
    lib = load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libctk_hip.so does not export {n}"
    assert set(names) == set(SYMBOLS), "ctypes binding and header disagree"
    assert lib.ctk_abi_version() == 6
    assert os.path.dirname(library_path()).endswith("control_toolkit_amd")   # in-tree, not site-packages


def test_ctypes_config_matches_c_struct(tmp_path):
    """sizeof/offsetof of ctk_config as the C compiler sees it == the ctypes mirror."""
    from control_toolkit_amd._capi import CtkConfig
    fields = [f[0] for f in CtkConfig._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "ctk_hip.h"\nint main(){printf("%zu", sizeof(ctk_config));'
    for f in fields:
        prog += f'printf(" %zu", offsetof(ctk_config, {f}));'
    prog += "return 0;}"
    c = tmp_path / "s.c"; c.write_text(prog)
    exe = tmp_path / "s"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(CtkConfig)
    for name, off in zip(fields, out[1:]):
        assert getattr(CtkConfig, name).offset == int(off), name


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from control_toolkit_amd import CtkEngine, CtkError
    with pytest.raises(CtkError, match="no HIP device|No HIP|no CPU fallback"):
        CtkEngine("mppi", "ODE", num_rollouts=32, mpc_horizon=10, dt=0.02)
    with pytest.raises(ValueError):
        CtkEngine("nope", "ODE", num_rollouts=32, mpc_horizon=10, dt=0.02)
    with pytest.raises(NotImplementedError):
        CtkEngine("mppi", "LSTM-6IN-32H1-32H2-5OUT-0", num_rollouts=32, mpc_horizon=10, dt=0.02)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under control_toolkit_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "control_toolkit_amd")
    pat = re.compile(r"^\s*(from|import)\s+\S*oracle|importlib.*oracle|CDLL\(.*oracle", re.I)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                for line in open(os.path.join(dp, f)):
                    assert not pat.search(line), f"{f}: {line.strip()}"
    # bench.py: only inside the cpu_baseline legs (functions named cpu_baseline*)
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    inside, total = 0, 0
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for n in ast.walk(fn):
            if (isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle") or \
               (isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)):
                assert fn.name.startswith("cpu_baseline"), f"bench.py:{n.lineno} imports the oracle inside {fn.name}()"
                inside += 1
    for n in ast.walk(tree):
        if (isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle") or \
           (isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)):
            total += 1
    assert inside == total >= 3            # no module-level import of the oracle either


def test_hip_library_gate_and_optimizer_discovery():
    from control_toolkit_amd.computation_library import HipLibrary
    from control_toolkit_amd.others.globals_and_utils import import_optimizer_by_name, find_optimizer_if_it_exists, create_rng
    from control_toolkit_amd.Predictors import PredictorWrapper
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
    lib = HipLibrary()
    assert lib.lib == "HIP" and lib.to_tensor([1, 2], lib.float32).dtype == np.float32
    lib.set_device("gpu:0")(lambda: None)
    with pytest.raises(ValueError):
        lib.set_device("/device:CPU:0")
    # discovery by file name, class name == file stem (reference globals_and_utils.py:103-133)
    for name in ("mppi-hip", "cem-hip", "rpgd-hip", "random-action-hip"):
        cls = import_optimizer_by_name(name)
        assert cls.__name__ == "optimizer_" + name.replace("-", "_")
        assert find_optimizer_if_it_exists(name)[0] == cls.__name__
    with pytest.raises(ValueError):
        import_optimizer_by_name("does-not-exist")
    Mppi = import_optimizer_by_name("mppi-hip")
    lim = (np.array([-1.0], np.float32), np.array([1.0], np.float32))
    kw = dict(predictor=PredictorWrapper(), cost_function=CostFunctionWrapper(), control_limits=lim, seed=1,
              cc_weight=1.0, R=1.0, LBD=100.0, mpc_horizon=10, num_rollouts=32, NU=1000.0, SQRTRHOINV=0.03,
              period_interpolation_inducing_points=1, optimizer_logging=False, calculate_optimal_trajectory=False,
              mpc_timestep=0.02)   # extra YAML keys are swallowed by **kwargs like the reference's
    with pytest.raises(ValueError, match="does not support"):   # reference Optimizers/__init__.py:27-28
        Mppi(computation_library=object(), **kw)
    o = Mppi(computation_library=lib, **kw)
    assert o.optimizer_name == "mppi-hip" and o.num_rollouts == 32 and o.mpc_horizon == 10 and o.logging_values == {}
    assert create_rng("x", None, lib).seed > 0                       # None -> datetime seed (:87-91)
    r = create_rng("x", 3, lib, mode="host")
    assert r.normal([4, 2, 1]).shape == (4, 2, 1) and 0 <= r.uniform([3]).min() < 1
    with pytest.raises(NotImplementedError):
        PredictorWrapper().configure(batch_size=1, dt=0.02, predictor_specification="LSTM-6IN-32H1-32H2-5OUT-0")
    with pytest.raises(ValueError):
        PredictorWrapper().configure(batch_size=1, dt=0.02, predictor_specification="GRU-5IN-32H1-32H2-4OUT-0")   # no weights
    g = PredictorWrapper(weights=np.zeros(10212, np.float32))
    g.configure(batch_size=1, dt=0.02, predictor_specification="GRU-5IN-32H1-32H2-4OUT-0")
    assert g.kind == "GRU"
    with pytest.raises(ValueError):
        PredictorWrapper().configure(batch_size=1, dt=0.02, predictor_specification="MLP")   # no weights


def test_controller_config_errors_match_reference():
    from control_toolkit_amd.Controllers.controller_mpc import controller_mpc
    lim = (np.array([-1.0], np.float32), np.array([1.0], np.float32))
    with pytest.raises(ValueError, match="could not be interpreted"):   # reference Controllers/__init__.py:57-58
        controller_mpc("CartPole", lim, {}, config_controllers={"mpc": {"computation_library": "tensorflow", "controller_logging": False}})
    c = controller_mpc("CartPole", lim, {"target_position": 0.1},
                       config_controllers={"mpc": {"optimizer": "mppi-hip", "computation_library": "hip", "controller_logging": True,
                                                   "device": "gpu:0"}},
                       config_optimizers={})
    assert c.controller_name == "mpc" and c.has_optimizer and c.lib.lib == "HIP"
    assert float(c.variable_parameters.target_position) == np.float32(0.1)
    c.update_logs({"Q_logged": np.zeros((2, 3, 1), np.float32), "J_logged": np.zeros(2, np.float32)})
    c.update_logs({"Q_logged": np.ones((2, 3, 1), np.float32), "J_logged": np.ones(2, np.float32)})
    out = c.get_outputs()                                               # stacked along axis 0 (:159-168)
    assert out["Q_logged"].shape == (2, 2, 3, 1) and out["s_logged"] is None


def test_cost_yaml_hot_reload_and_declarative_predictor(tmp_path):
    """reference cost_function_wrapper.py:56-74 + CostFunctionUpdater.py: YAML section -> parameters, a watcher
    raises the flag on modification, the controller consumes it at the top of step()."""
    import time
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper, CostFunctionUpdater, DEFAULT_COST
    from control_toolkit_amd.Predictors import PredictorWrapper
    y = tmp_path / "config_cost_function.yml"
    y.write_text("cost_function_name_default: quadratic-boundary\nCartPole:\n  quadratic_boundary:\n    dd_weight: 100.0\n    R: 2.0\n"
                 "  other:\n    ep_weight: 1.0\n")
    c = CostFunctionWrapper(config_path=str(y), watch=False)
    c.configure(8, 5, environment_name="CartPole")
    assert c.cost_function_name == "quadratic_boundary"                  # default name, '-' -> '_' (reference :76-88)
    assert c.parameters["dd_weight"] == 100.0 and c.parameters["R"] == 2.0 and c.parameters["ep_weight"] == DEFAULT_COST["ep_weight"]
    assert str(y) in CostFunctionUpdater.active_watchers
    assert not c.cost_function_updater.poll_now()                        # unchanged file: no flag
    time.sleep(0.01)
    y.write_text("cost_function_name_default: quadratic-boundary\nCartPole:\n  quadratic_boundary:\n    dd_weight: 7.5\n    R: 2.0\n")
    assert c.cost_function_updater.poll_now() and c.reload_cost_parameters_from_config_flag
    assert c.parameters["dd_weight"] == 100.0                            # nothing changes until the controller consumes the flag
    c.update_cost_parameters_from_config()
    assert c.parameters["dd_weight"] == 7.5 and c.version == 1 and not c.reload_cost_parameters_from_config_flag
    # a second wrapper on the same path replaces the first watcher (reference :22-24); explicit spec selects the section
    c2 = CostFunctionWrapper(config_path=str(y), watch=True)
    y.write_text("CartPole:\n  other:\n    ep_weight: 1.0\n")
    c2.configure(8, 5, environment_name="CartPole", cost_function_specification="other")
    assert CostFunctionUpdater.active_watchers[str(y)] is c2.cost_function_updater and c2.parameters["ep_weight"] == 1.0
    time.sleep(0.01)
    y.write_text("CartPole:\n  other:\n    ep_weight: 3.0\n")
    t0 = time.time()
    while not c2.reload_cost_parameters_from_config_flag and time.time() - t0 < 5:
        time.sleep(0.02)                                                  # the watcher thread, not poll_now
    assert c2.reload_cost_parameters_from_config_flag
    c2.update_cost_parameters_from_config()
    assert c2.parameters["ep_weight"] == 3.0
    CostFunctionUpdater.stop_all_watchers()
    assert CostFunctionUpdater.active_watchers == {}
    with pytest.raises(ValueError, match="unknown cost parameters"):
        bad = tmp_path / "bad.yml"; bad.write_text("CartPole:\n  default:\n    not_a_term: 1.0\n")
        CostFunctionWrapper(config_path=str(bad), watch=False).configure(8, 5, environment_name="CartPole")
    with pytest.raises(NotImplementedError, match="not built"):
        CostFunctionWrapper(config_path=str(y), watch=False).configure(8, 5, environment_name="Acrobot")
    with pytest.raises(KeyError):      # a built environment the file has no section for
        CostFunctionWrapper(config_path=str(y), watch=False).configure(8, 5, environment_name="Quad2D")
    with pytest.raises(FileNotFoundError):
        CostFunctionWrapper(config_path=str(tmp_path / "missing.yml"), watch=False).configure(8, 5, environment_name="CartPole")
    # declarative predictor: YAML -> dynamics constants + weights file
    np.save(tmp_path / "net.npy", np.arange(1380, dtype=np.float32))
    p = tmp_path / "predictor.yml"
    p.write_text("CartPole:\n  dynamics: {m_pole: 0.1, L: 0.25}\n  intermediate_steps: 2\n  weights_file: net.npy\n")
    pw = PredictorWrapper.from_yaml(str(p), "CartPole")
    assert pw.parameters["m_pole"] == 0.1 and pw.parameters["g"] == 9.81 and pw.intermediate_steps == 2 and pw.weights.size == 1380
    pw.configure(batch_size=4, dt=0.02, predictor_specification="MLP")
    assert pw.kind == "MLP"
    with pytest.raises(ValueError):
        q = tmp_path / "q.yml"; q.write_text("dynamics: {}\nfoo: 1\n")
        PredictorWrapper.from_yaml(str(q))


def test_environment_tables_agree_between_library_host_mirror_and_oracle():
    """ctk_env_info / ctk_param_name (no GPU needed) == the static host tables == the oracle's parameter lists."""
    from control_toolkit_amd._capi import environment_info, ENVIRONMENTS
    from control_toolkit_amd.Predictors import ENVIRONMENT_DIMS, DEFAULT_DYNAMICS_BY_ENV, network_weight_count
    from control_toolkit_amd.Cost_Functions import DEFAULT_COST_BY_ENV, DEFAULT_ATTRIBUTES_BY_ENV
    from oracle import ctk_oracle as O
    assert set(ENVIRONMENTS) == set(ENVIRONMENT_DIMS) == set(O.ENVIRONMENTS)
    for name, cls in O.ENVIRONMENTS.items():
        S, C, params = environment_info(name)
        assert (S, C) == ENVIRONMENT_DIMS[name] == (cls.S, cls.C)
        assert params == cls().param_names()
        host = dict(DEFAULT_DYNAMICS_BY_ENV[name], **DEFAULT_ATTRIBUTES_BY_ENV[name], **DEFAULT_COST_BY_ENV[name])
        assert set(host) == set(params)
        for k in params:
            assert host[k] == getattr(cls(), k), (name, k)
    assert network_weight_count("MLP", 4, 1) == 1380 == O.mlp_num_weights() and network_weight_count("GRU", 4, 1) == 10212 == O.gru_num_weights()
    assert network_weight_count("MLP", 6, 2) == O.mlp_num_weights(8, 6)
    with pytest.raises(NotImplementedError):
        environment_info("Acrobot")


def test_network_name_convention_is_parsed_and_checked():
    """reference config_controllers.yml:8 `GRU-6IN-32H1-32H2-5OUT-0`; controller_mpc.py:67-73 hands the name to the predictor"""
    from control_toolkit_amd.Predictors import parse_predictor_specification as parse, check_network_sizes, network_weight_count, PredictorWrapper
    assert parse(None) == ("ODE", None) and parse("ODE") == ("ODE", None) and parse("MLP") == ("MLP", None)
    assert parse("GRU-6IN-32H1-32H2-5OUT-0") == ("GRU", dict(inputs=6, h1=32, h2=32, outputs=5))
    assert parse("Dense-5IN-16H1-24H2-4OUT-3") == ("MLP", dict(inputs=5, h1=16, h2=24, outputs=4))
    with pytest.raises(NotImplementedError):
        parse("LSTM-5IN-16H1-16H2-4OUT-0")
    with pytest.raises(NotImplementedError, match="hidden layers"):
        parse("Dense-5IN-16H1-16H2-16H3-4OUT-0")
    assert check_network_sizes("x", None, 4, 1) == (32, 32)
    assert check_network_sizes("Dense-5IN-16H1-24H2-4OUT-3", parse("Dense-5IN-16H1-24H2-4OUT-3")[1], 4, 1) == (16, 24)
    with pytest.raises(ValueError, match="6 inputs"):
        check_network_sizes("GRU-6IN-32H1-32H2-5OUT-0", parse("GRU-6IN-32H1-32H2-5OUT-0")[1], 4, 1)
    assert check_network_sizes("Dense-5IN-64H1-48H2-4OUT-0", parse("Dense-5IN-64H1-48H2-4OUT-0")[1], 4, 1) == (64, 48)    # MLPs up to 64 / 64
    with pytest.raises(NotImplementedError, match="128 / 64"):
        check_network_sizes("Dense-5IN-128H1-64H2-4OUT-0", parse("Dense-5IN-128H1-64H2-4OUT-0")[1], 4, 1)
    with pytest.raises(NotImplementedError, match="GRU predictor kernels hold up to 32"):
        check_network_sizes("GRU-5IN-64H1-64H2-4OUT-0", parse("GRU-5IN-64H1-64H2-4OUT-0")[1], 4, 1)
    assert network_weight_count("MLP", 4, 1) == 1380 and network_weight_count("GRU", 4, 1) == 10212
    assert network_weight_count("MLP", 4, 1, (16, 24)) == 5 * 16 + 16 + 16 * 24 + 24 + 24 * 4 + 4
    p = PredictorWrapper(weights=np.zeros(network_weight_count("MLP", 4, 1, (16, 24)), np.float32))
    p.configure(batch_size=8, dt=0.02, predictor_specification="Dense-5IN-16H1-24H2-4OUT-0")
    assert p.kind == "MLP" and p.hidden_sizes == (16, 24)
    with pytest.raises(ValueError, match="expects"):
        p.configure(batch_size=8, dt=0.02, predictor_specification="Dense-5IN-16H1-16H2-4OUT-0")


def test_user_environment_library_builds_and_describes_itself():
    """a plant + cost that is not in csrc/: tests/envs/pendulum_env.h compiled by control_toolkit_amd/build_env.py into a library of its own
    (hipcc cross-compiles without a GPU; cached by content, so only the first run pays the ~1 minute).  No compute calls here."""
    import os
    from control_toolkit_amd import _capi
    from control_toolkit_amd.build_env import build_environment, environment_name_of, register_environment
    header = os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs", "pendulum_env.h")
    assert environment_name_of(header) == "Pendulum"
    name, lib = build_environment(header)
    assert name == "Pendulum" and os.path.exists(lib) and "_env_builds" in lib
    assert build_environment(header)[1] == lib                         # cache hit: same content, same library
    assert register_environment(header) == "Pendulum"
    S, C, names = _capi.environment_info("Pendulum")
    assert (S, C) == (2, 1) and names[0] == "g" and names[-1] == "terminal_weight" and len(names) == 11
    assert _capi.environment_defaults("Pendulum")["torque_gain"] == 12.0
    ulib, eid = _capi.environment_library("Pendulum")
    assert eid == 3 and ulib.ctk_abi_version() == _capi.load_library().ctk_abi_version()
    for sym in _capi.SYMBOLS:                                          # the user library is the whole engine: every declared symbol
        assert hasattr(ulib, sym), sym
    assert _capi.load_library().ctk_environment_name(3) is None        # the product library has no fourth environment
    with pytest.raises(NotImplementedError, match="register_environment"):
        _capi.environment_info("Acrobot")
    # the host-side wrappers resolve the registered name
    from control_toolkit_amd.Predictors import PredictorWrapper
    from control_toolkit_amd.Cost_Functions import CostFunctionWrapper
    assert PredictorWrapper(environment_name="Pendulum").num_states == 2
    cw = CostFunctionWrapper({"ang_weight": 70.0}, watch=False, environment_name="Pendulum")
    assert cw.parameters["ang_weight"] == 70.0 and cw.parameters["damping"] == pytest.approx(0.1)
    with pytest.raises(ValueError, match="unknown cost parameters"):
        CostFunctionWrapper({"dd_weight": 1.0}, watch=False, environment_name="Pendulum")


def test_rpgd_one_launch_forward_loop_never_waits_for_its_own_stores(tmp_path):
    """A code-generation guard (no GPU needed: the built library is disassembled).  The forward pass of ctk_rpgd_mlp_persistent publishes one
    8-byte word per lane and step and must never wait for those stores.  Twice while the kernel was written the compiler put an
    `s_waitcnt vmcnt(0)` inside that loop — for a pointer reloaded from scratch, then for a weight whose load it had not yet waited for —
    and each step then waited for the previous step's store: 1 156 us per MPC step instead of 720, with no spill in the loop and nothing
    in a profile to point at it.  The operands are pinned in front of the loop (`mlp_pin`); this test keeps it that way."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "control_toolkit_amd", "libctk_hip.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("llvm-objdump or the built library is not here")
    work = str(tmp_path / "libctk_hip.so")
    shutil.copy(lib, work)
    subprocess.run([objdump, "--offloading", "libctk_hip.so"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)   # bundles land beside the file
    listing = None
    for name in sorted(os.listdir(str(tmp_path))):
        if name.endswith("gfx950"):
            text = subprocess.run([objdump, "-d", name], cwd=str(tmp_path), capture_output=True, text=True, timeout=300).stdout
            if "<_Z23ctk_rpgd_mlp_persistent" in text:
                listing = text.splitlines()
    assert listing is not None, "ctk_rpgd_mlp_persistent is not in the library"
    start = next(i for i, l in enumerate(listing) if re.match(r"^[0-9a-f]+ <_Z23ctk_rpgd_mlp_persistent.*>:", l))
    end = next((i for i in range(start + 1, len(listing)) if re.match(r"^[0-9a-f]+ <.*>:", listing[i])), len(listing))
    body = listing[start:end]
    stores = [i for i, l in enumerate(body) if "global_store_dwordx2" in l and " sc1" in l and "sc0" not in l]
    assert len(stores) == 1, f"expected ONE publish store (agent-scope 8-byte word) in the kernel, found {len(stores)}"
    s = stores[0]
    back = next(i for i in range(s, len(body)) if "s_cbranch_scc1" in body[i] or "s_cbranch_scc0" in body[i])     # the step loop's back edge
    loop = body[max(0, s - 12):back + 1]
    assert sum("v_mfma_f32_16x16x4" in l for l in loop) >= 9, "this is not the forward loop (a step is 9 wide + 4 narrow matrix products)"
    waits = [l.strip() for l in loop if "s_waitcnt" in l and "vmcnt" in l]
    spills = [l.strip() for l in loop if "scratch_" in l]
    assert not waits, f"the forward loop waits for memory (and so for its own stores): {waits}"
    assert not spills, f"the forward loop spills: {spills}"
