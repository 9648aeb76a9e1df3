"""Run the REFERENCE's own, unmodified Python callers on the HIP backend.

    python tools/reference_on_hip.py stage        # here (needs /root/reference): copy the reference's Python package
                                                  # into the UNTRACKED scratch directory .refstage/ and add the
                                                  # "hip" backend to its curobolib/backends/__init__.py
    python tools/reference_on_hip.py run          # on the GPU box (gpurun ships .refstage/ with the snapshot)
    python tools/reference_on_hip.py clean        # remove .refstage/ again (nothing of the reference stays in the repo)

`run` does two things and writes gpurun_out/ref_on_hip/{report.json,*.log}:

  A. pytest of the reference's own test files that exercise the optimiser / kernel-wrapper layer
     (tests/_src/optim/..., tests/_src/curobolib/cuda_ops) -- unmodified files, unmodified
     cuda_ops / optim sources, kernels = libcurobo_hip.so through curobo_amd.backends;
  B. the reference's autograd wrappers (cuda_ops.KinematicsFusedFunction, SelfCollisionDistance,
     BSplineIdxKernel, LBFGScu, wolfe_line_search) called next to this repository's twins
     (curobo_amd.hip_ops) on the same inputs: outputs must be BIT-identical (both end in the same
     C-ABI entry points), which shows the twins add nothing the reference's files do not do.

The only edit to the staged reference is the backend hook of INTEGRATION.md (added lines only).
"""

import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, ".refstage")
REF = "/root/reference/curobo"
OUT = os.path.join(ROOT, "gpurun_out", "ref_on_hip")

HOOK_FN = '''

def _try_load_hip():
    """MI355X / ROCm backend: ctypes shims over libcurobo_hip.so (include/curobo_hip.h)."""
    import torch

    if not (torch.cuda.is_available() and torch.version.hip):
        return None
    from curobo_amd import backends as hip  # raises ImportError if the .so is missing

    return hip.get_backend()

'''
HOOK_SELECT = '''
    if backend_pref in ("auto", "hip"):
        hip_modules = _try_load_hip()
        if hip_modules is not None:
            return ("hip", hip_modules)
        if backend_pref == "hip":
            raise RuntimeError("kernel_backend='hip' requested but no ROCm device / libcurobo_hip.so")
'''

REFERENCE_TESTS = [
    "tests/_src/optim/gradient/test_lbfgs.py",
    "tests/_src/optim/gradient/test_lsr1.py",
    "tests/_src/optim/gradient/test_conjugate_gradient.py",
    "tests/_src/optim/test_optimizer_cuda_graph.py",
    "tests/_src/optim/test_multi_stage_optimizer.py",
    "tests/_src/transition/test_fns_state_transition.py",
    "tests/_src/curobolib/cuda_ops",
]


def stage():
    if os.path.exists(STAGE):
        shutil.rmtree(STAGE)
    keep = (".py", ".yml", ".yaml", ".json", ".cuh", ".h", ".cu", ".cpp", ".toml", ".txt")

    def ignore(d, names):
        out = []
        for n in names:
            p = os.path.join(d, n)
            if os.path.isdir(p):
                if n in ("__pycache__", "assets", "perception", ".git"):
                    out.append(n)
            elif not n.endswith(keep):
                out.append(n)
        return out
    shutil.copytree(REF, os.path.join(STAGE, "curobo"), ignore=ignore)
    init = os.path.join(STAGE, "curobo", "_src", "curobolib", "backends", "__init__.py")
    s = open(init).read()
    a1 = "def _auto_select_backend() -> tuple:"
    a2 = "    backend_pref = runtime.kernel_backend.lower()\n"
    assert a1 in s and a2 in s
    s = s.replace(a1, HOOK_FN.lstrip("\n") + "\n" + a1, 1).replace(a2, a2 + HOOK_SELECT, 1)
    open(init, "w").write(s)
    # ---- the Warp side: `warp` = the pure-Python stand-in of tests/golden/warp_emulator + the device shim,
    # `yourdfpy` = the xml.etree stand-in, the pytest plugin that installs the hooks of INTEGRATION.md at run time
    stubs = os.path.join(STAGE, "_stubs")
    src = os.path.join(ROOT, "tools", "refstage_stubs")
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "warp_emulator", "warp"), os.path.join(stubs, "warp"),
                    ignore=shutil.ignore_patterns("__pycache__"))
    with open(os.path.join(stubs, "warp", "__init__.py"), "a") as fh:
        fh.write(open(os.path.join(src, "warp_device_shim.py")).read())
    shutil.copytree(os.path.join(src, "yourdfpy"), os.path.join(stubs, "yourdfpy"))
    for f in ("hip_hooks.py", "refstage_plugin.py"):
        shutil.copy(os.path.join(src, f), os.path.join(stubs, f))
    # robot assets: the URDFs (text) of the robots the tests load; meshes are not needed (collision spheres come from the yaml)
    for robot in ("franka_description", "ur_description"):
        a = os.path.join(REF, "content", "assets", "robot", robot)
        if os.path.isdir(a):
            shutil.copytree(a, os.path.join(STAGE, "curobo", "content", "assets", "robot", robot),
                            ignore=lambda d, names: [n for n in names if os.path.isfile(os.path.join(d, n)) and not n.endswith((".urdf", ".xacro", ".yml"))])
    n = sum(len(f) for _, _, f in os.walk(STAGE))
    print(f"staged {n} files under {STAGE} (untracked scratch; `clean` removes it)")


def clean():
    shutil.rmtree(STAGE, ignore_errors=True)
    print("removed", STAGE)


def run_reference_tests(report):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([STAGE, ROOT, env.get("PYTHONPATH", "")])
    res = {}
    for t in REFERENCE_TESTS:
        path = os.path.join(STAGE, "curobo", t)
        log = os.path.join(OUT, t.replace("/", "_") + ".log")
        cmd = [sys.executable, "-m", "pytest", path, "-q", "-p", "no:cacheprovider", "--no-header", "-rN"]
        try:
            p = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
            out, rc = p.stdout + "\n---- stderr ----\n" + p.stderr[-4000:], p.returncode
        except subprocess.TimeoutExpired as e:
            out, rc = f"TIMEOUT\n{e.stdout}", -9
        open(log, "w").write(out)
        tail = [ln for ln in out.splitlines() if ln.strip()]
        summary = next((ln for ln in reversed(tail) if " passed" in ln or " failed" in ln or "error" in ln.lower()), tail[-1] if tail else "")
        res[t] = {"rc": rc, "summary": summary.strip()}
        print(t, "->", res[t])
    report["reference_pytest"] = res


WARP_SIDE_TESTS = [
    "tests/_src/cost/test_cost_scene_collision.py",
    "tests/_src/cost/test_cost_tool_pose.py",
    "tests/_src/cost/test_cost_cspace.py",
    "tests/_src/solver/test_solver_ik.py",
    "tests/_src/solver/seed_ik/test_seed_ik_solver.py",
    "tests/_src/rollout/test_rollout_robot.py",
    "tests/_src/collision/test_collision_robot_scene.py",
    "tests/_src/solver/test_solver_trajopt.py",
    "tests/_src/robot/kinematics/test_kinematics.py",
    "tests/_src/cost/test_cost_self_collision.py",
]


def run_warp_side_tests(report, tests=None):
    """the reference's own tests of its Warp-backed costs and of the IK solver, on the MI355X: `warp` is the stand-in
    (names + host emulation of set-up kernels), the hot-path autograd functions end in libcurobo_hip.so (hip_hooks.py)"""
    env = dict(os.environ)
    stubs = os.path.join(STAGE, "_stubs")
    env["PYTHONPATH"] = os.pathsep.join([stubs, STAGE, ROOT, env.get("PYTHONPATH", "")])
    res = {}
    for t in tests or WARP_SIDE_TESTS:
        path = os.path.join(STAGE, "curobo", t)
        tag = t.replace("/", "_")
        log = os.path.join(OUT, tag + ".log")
        env["REFSTAGE_REPORT"] = os.path.join(OUT, tag + ".hooks.json")
        cmd = [sys.executable, "-m", "pytest", path, "-q", "-p", "refstage_plugin", "-p", "no:cacheprovider", "--no-header", "-rf"] + (["-x"] if os.environ.get("REFSTAGE_X") else [])
        try:
            p = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=1500)
            out, rc, stdout = p.stdout + "\n---- stderr ----\n" + p.stderr[-6000:], p.returncode, p.stdout
        except subprocess.TimeoutExpired as e:
            out, rc, stdout = f"TIMEOUT\n{e.stdout}", -9, str(e.stdout)
        open(log, "w").write(out)
        tail = [ln for ln in stdout.splitlines() if ln.strip()]
        summary = next((ln for ln in reversed(tail) if " passed" in ln or " failed" in ln or " error" in ln), tail[-1] if tail else "")
        failed = [ln[len("FAILED "):].split(" - ")[0] for ln in tail if ln.startswith("FAILED ")]
        hooks = {}
        try:
            hooks = json.load(open(env["REFSTAGE_REPORT"]))
        except (OSError, ValueError):
            pass
        res[t] = {"rc": rc, "summary": summary.strip("= "), "failed": failed, **hooks}
        print(t, "->", res[t]["summary"], "| HIP entry points:", hooks.get("hip_entry_points_reached"))
    report["reference_warp_side_pytest"] = res


def compare_wrappers(report):
    """the reference's autograd wrappers vs curobo_amd.hip_ops on identical inputs, bit for bit"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, STAGE)  # the staged reference package must win over this repository's `curobo/` shim package
    for name in [m for m in sys.modules if m == "curobo" or m.startswith("curobo.")]:
        del sys.modules[name]
    import numpy as np
    import torch

    from curobo._src.curobolib import backends as ref_backends
    from curobo._src.curobolib.cuda_ops.geometry import SelfCollisionDistance as RefSelf
    from curobo._src.curobolib.cuda_ops.kinematics import KinematicsFusedFunction as RefKin
    from curobo._src.curobolib.cuda_ops.optimization import LBFGScu as RefLBFGS
    from curobo._src.curobolib.cuda_ops.trajectory import BSplineIdxKernel as RefBS

    from curobo_amd.hip_ops import BSplineIdxKernel, KinematicsFusedFunction, LBFGScu, SelfCollisionDistance
    from curobo_amd.kinematics import KinematicsCfg

    dev = torch.device("cuda:0")
    res = {"backend_selected_by_reference": ref_backends.get_backend_name()}
    rng = np.random.default_rng(0)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dt)  # noqa: E731

    # ---- FK forward + backward (franka, spheres + jacobian-free path and the Jacobian path)
    kcfg = KinematicsCfg.from_packaged("franka", device=dev)
    k = kcfg.kinematics_config
    B, H, D = 6, 5, k.num_dof
    lo, hi = k.joint_limits_position[0].cpu().numpy(), k.joint_limits_position[1].cpu().numpy()
    qn = (lo + (hi - lo) * rng.uniform(size=(B, H, D))).astype(np.float32)
    gs = rng.normal(size=(B, H, k.num_spheres, 4)).astype(np.float32)
    gp = rng.normal(size=(B, H, k.num_pose_links, 3)).astype(np.float32)
    gq = rng.normal(size=(B, H, k.num_pose_links, 4)).astype(np.float32)
    for jac in (False, True):
        outs = []
        for F in (RefKin, KinematicsFusedFunction):
            buf = KinematicsFusedFunction.create_buffers(B, H, k)
            q = t(qn).requires_grad_(True)
            env = torch.zeros(B, dtype=torch.int32, device=dev)
            o = F.apply(q, buf["batch_link_position"], buf["batch_link_quaternion"], buf["batch_robot_spheres"], buf["batch_com"],
                        buf["batch_jacobian"], buf["batch_cumul_mat"], k, buf["grad_out_q"], buf["grad_out_q_jacobian"],
                        buf["grad_in_link_pos"], buf["grad_in_link_quat"], buf["grad_in_robot_spheres"], buf["grad_in_com"],
                        jac, True, False, env, H)
            pos, quat, sph = o[0], o[1], o[2]
            ((sph * t(gs)).sum() + (pos * t(gp)).sum() + (quat * t(gq)).sum()).backward()
            torch.cuda.synchronize()
            outs.append([x.detach().clone() for x in (pos, quat, sph, o[4], buf["batch_cumul_mat"], q.grad)])
        res[f"KinematicsFusedFunction(jacobian={jac})"] = all(torch.equal(a, b) for a, b in zip(*outs))

    # ---- self collision
    S = k.num_spheres
    sc = k.self_collision
    sph = outs[0][2].reshape(B, H, S, 4).contiguous()
    vals = []
    for F in (RefSelf, SelfCollisionDistance):
        x = sph.clone().requires_grad_(True)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)  # noqa: E731
        d = F.apply(x, z(B, H, 1), z(B, H, S, 4), z(1), z(B, H, S, dt=torch.uint8), t([7.0]), sc.sphere_padding, sc.collision_pairs,
                    z(1), z(2, dt=torch.int16), 1, 256, False, False)  # return_loss=False: the reference's default (use_grad_input);
        # its True branch multiplies [B,H,S,4] by [B,H,1] and cannot broadcast (cuda_ops/geometry.py:101)
        d.sum().backward()
        torch.cuda.synchronize()
        vals.append((d.detach().clone(), x.grad.clone()))
    res["SelfCollisionDistance"] = torch.equal(vals[0][0], vals[1][0]) and torch.equal(vals[0][1], vals[1][1])
    res["SelfCollisionDistance_hits"] = int((vals[0][0] > 0).sum())

    # ---- B-spline forward + backward
    b, nk, deg, interp = 6, 12, 3, 2
    ph = (nk + deg + 1) * interp + 1
    un = rng.normal(size=(b, nk, D)).astype(np.float32)
    gpp = rng.normal(size=(b, ph, D)).astype(np.float32)
    vals = []
    for F in (RefBS, BSplineIdxKernel):
        u = t(un).requires_grad_(True)
        z1 = lambda: torch.zeros(1, D, device=dev)  # noqa: E731
        o4 = [torch.zeros(b, ph, D, device=dev) for _ in range(4)]
        idx = torch.zeros(b, dtype=torch.int32, device=dev)
        p, v, a_, j = F.apply(u, z1(), z1(), z1(), z1(), z1(), z1(), z1(), z1(), idx, idx, *o4, torch.zeros(b, device=dev), t([0.05]),
                              torch.zeros(1, dtype=torch.uint8, device=dev), torch.zeros(b, nk, D, device=dev), deg)
        ((p * t(gpp)).sum() + 0.1 * (v * t(gpp)).sum() + 0.01 * (a_ * t(gpp)).sum()).backward()
        torch.cuda.synchronize()
        vals.append([x.detach().clone() for x in (p, v, a_, j, u.grad)])
    res["BSplineIdxKernel"] = all(torch.equal(a, b_) for a, b_ in zip(*vals))

    # ---- L-BFGS step
    m, V, nb = 7, 84, 16
    mk = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
    st = dict(step=np.zeros((nb, V), np.float32), rho=np.abs(mk(m, nb)) * 0.1, y=mk(m, nb, V), s=mk(m, nb, V), q=mk(nb, V), g=mk(nb, V),
              x0=mk(nb, V), g0=mk(nb, V))
    vals = []
    for F in (RefLBFGS, LBFGScu):
        dv = {kk: t(a.copy()) for kk, a in st.items()}
        out = F.apply(dv["step"], dv["rho"].view(m, nb, 1, 1), dv["y"].view(m, nb, V, 1), dv["s"].view(m, nb, V, 1), dv["q"],
                      dv["g"].view(nb, 1, V), dv["x0"].view(nb, V, 1), dv["g0"].view(nb, V, 1), 0.01, True, True)
        torch.cuda.synchronize()
        vals.append([out.clone(), dv["rho"].clone(), dv["y"].clone(), dv["s"].clone(), dv["x0"].clone(), dv["g0"].clone()])
    res["LBFGScu"] = all(torch.equal(a, b_) for a, b_ in zip(*vals))
    report["wrappers_bit_identical"] = res
    print(json.dumps(res, indent=1))


def run():
    os.makedirs(OUT, exist_ok=True)
    if not os.path.isdir(os.path.join(STAGE, "curobo")):
        raise SystemExit(".refstage/ is missing: run `python tools/reference_on_hip.py stage` in the container first")
    report = {"what": "the reference's unmodified Python callers over curobo_amd.backends (libcurobo_hip.so) on MI355X"}
    if "warp" in sys.argv[2:]:  # only the Warp-side part (iterating on the hooks)
        run_warp_side_tests(report, [a for a in sys.argv[3:] if a.endswith(".py")] or None)
        json.dump(report, open(os.path.join(OUT, "report_warp_side.json"), "w"), indent=1)
        return
    run_reference_tests(report)
    run_warp_side_tests(report)
    try:
        compare_wrappers(report)
    except Exception as e:  # noqa: BLE001
        import traceback

        report["wrappers_bit_identical"] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-3000:]}
        print(report["wrappers_bit_identical"]["trace"])
    json.dump(report, open(os.path.join(OUT, "report.json"), "w"), indent=1)
    print(json.dumps(report["reference_pytest"], indent=1))


if __name__ == "__main__":
    {"stage": stage, "run": run, "clean": clean}[sys.argv[1] if len(sys.argv) > 1 else "run"]()
