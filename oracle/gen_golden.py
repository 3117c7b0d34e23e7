#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the unmodified reference.

Test infrastructure.  Runs only in the build container, where the reference is
mounted read-only at /root/reference; it imports RT_bench.py from there (nothing
of it is copied) and stores inputs + expected outputs as small .npz fixtures.
Refuses to run if the reference is absent (e.g. on the GPU box).

Recipe: after import, set the module globals that RT_bench.py's
`__main__` block would set (`f`, `gamma`), build the field with genZ +
interpolacion, then call trazar / opN / n_gradient directly.  Non-preset ray
batches are obtained by swapping `R.constants` for a function that returns a
modified 13-tuple (trazar looks `constants` up as a module global, :793).

Usage: MPLBACKEND=Agg python3 oracle/gen_golden.py [--only field,step,traj_vert,...]
"""
import argparse
import os
import sys
import time

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
STRIDE = 64


def load_reference():
    if not os.path.isfile(os.path.join(REF, "RT_bench.py")):
        sys.exit("gen_golden: /root/reference/RT_bench.py not present; fixtures can only be made in the build container")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import RT_bench as R  # noqa
    return R


SCEN = {"interface": ("1", "interface", 1), "fisheye": ("2", "fisheye", 1),
        "vert_heterogeneous": ("3", "vert_heterogeneous", 1), "anisotropy": ("4", "vert_heterogeneous", 3)}


def build_field(R, scen):
    choice, fname, gamma = SCEN[scen]
    R.f = getattr(R, fname)
    R.gamma = gamma
    c = R.constants(choice)
    linx, liny, X, Y, Z = R.genZ(*c[5:9])
    z, grd, _ = R.interpolacion(linx, liny, Z, X, Y)
    return c, linx, liny, Z, z, grd


def sub_rows(s_ray, d_ray):
    """every STRIDE-th row + the last 3 written rows per ray"""
    R_ = s_ray.shape[2]
    strided = s_ray[::STRIDE].copy()
    last = np.zeros((3, 6, R_))
    for k in range(R_):
        i = int(d_ray[2, k])
        for j in range(3):
            last[j, :, k] = s_ray[max(i - 2 + j, 0), :, k]
    return strided, last


def gen_field(R):
    rng = np.random.default_rng(0)
    for scen in ("interface", "fisheye", "vert_heterogeneous"):
        c, linx, liny, Z, z, grd = build_field(R, scen)
        xi, xs, yi, ys = c[5:9]
        px = rng.uniform(xi, xs, 1024)
        py = rng.uniform(yi, ys, 1024)
        # a few points on/near grid nodes, box corners and just outside the box (rays overshoot by <= 1 step)
        px[:8] = [xi, xs, xi, xs, linx[10], linx[11], xs + 0.002, xi - 0.002]
        py[:8] = [yi, ys, ys, yi, liny[10], liny[12], ys + 0.002, yi - 0.002]
        out = np.array([np.concatenate(([R.n_gradient(np.array((a, b)), grd, z)[0]],
                                        R.n_gradient(np.array((a, b)), grd, z)[1])) for a, b in zip(px, py)])
        qy, qx = Z.shape
        cdy = grd[0].get_coeffs().reshape(qy, qx)
        cdx = grd[1].get_coeffs().reshape(qy, qx)
        blocks = {}
        for name, arr in (("Z", Z), ("cdy", cdy), ("cdx", cdx)):
            blocks[name + "_c00"] = arr[:8, :8].copy()
            blocks[name + "_c11"] = arr[-8:, -8:].copy()
            blocks[name + "_mid"] = arr[qy // 2:qy // 2 + 8, qx // 2:qx // 2 + 8].copy()
        np.savez_compressed(os.path.join(OUT, f"field_{scen}.npz"), limits=np.array(c[5:9], float),
                            delta=R.DELTA, qx=qx, qy=qy, x_head=linx[:4], x_tail=linx[-4:], y_head=liny[:4],
                            y_tail=liny[-4:], tx=grd[0].get_knots()[1], ty=grd[0].get_knots()[0],
                            px=px, py=py, n=out[:, 0], gx=out[:, 1], gy=out[:, 2], **blocks)
        print("field", scen, Z.shape)


def gen_step(R):
    """64 random states per method, one opN call each (vert grid; gamma=1 for op1-9, 3 for op10/11)."""
    rng = np.random.default_rng(1)
    c, linx, liny, Z, z, grd = build_field(R, "vert_heterogeneous")
    step = R.DELTA_S
    res = {}
    for m in range(1, 12):
        gamma = 3 if m >= 10 else 1
        R.gamma = gamma
        op = getattr(R, f"op{m}")
        st = np.zeros((64, 7)); hist = np.zeros((64, 6)); out = np.zeros((64, 6))
        for q in range(64):
            x, y = rng.uniform(-2, 5), rng.uniform(-2.5, 1)
            th = rng.uniform(-np.pi, np.pi)
            pos = np.array((x, y))
            n, g = R.n_gradient(pos, grd, z)
            coef = R.anisotropy(th, gamma)
            u = np.array((np.cos(th), np.sin(th)))
            # op7: three previous positions ending at pos, roughly along -u
            h = [pos - 2 * step * u + rng.normal(0, 1e-5, 2), pos - step * u + rng.normal(0, 1e-5, 2), pos]
            if m == 7:
                R.VECTOR_LIST.clear(); R.VECTOR_LIST.extend([a.copy() for a in h])
            fp, fa, fn, fg = op(th, n, g, u, pos, coef, grd, z, step)
            R.VECTOR_LIST.clear()
            st[q] = (x, y, th, n, g[0], g[1], coef)
            hist[q] = np.concatenate(h)
            out[q] = (fp[0], fp[1], fa, fn, fg[0], fg[1])
        res[f"st{m}"] = st; res[f"hist{m}"] = hist; res[f"out{m}"] = out
    R.gamma = 1
    np.savez_compressed(os.path.join(OUT, "step_methods.npz"), step=step, **res)
    print("step fixtures done")


def run_traj(R, scen, method, step, divisor, rays=None):
    """rays: None -> preset, else (theta_v, pos_x) replacing the preset (ray_count follows)."""
    c, linx, liny, Z, z, grd = build_field(R, scen)
    choice = SCEN[scen][0]
    orig = R.constants
    if rays is not None:
        theta_v, pos_x = rays
        cl = list(c); cl[1] = len(theta_v); cl[2] = np.asarray(theta_v); cl[3] = np.asarray(pos_x)
        R.constants = lambda uc, _t=tuple(cl): _t
        c = tuple(cl)
    try:
        t = time.time()
        s_ray, d_ray, _, errors = R.trazar(getattr(R, f"op{method}"), z, grd, False, step, divisor, choice)
        dt = time.time() - t
    finally:
        R.constants = orig
    strided, last = sub_rows(s_ray, d_ray)
    print(f"traj {scen} op{method}: rays={c[1]} steps={int(d_ray[2].sum())} {dt:.1f}s")
    return dict(theta=np.asarray(c[2], float)[:c[1]], pos_x=np.asarray(c[3], float), step=step, divisor=divisor,
                gamma=c[0], s_max=float(c[4]), box=np.array(c[5:9], float), max_size=s_ray.shape[0],
                d_ray=d_ray, errors=errors, strided=strided, last=last, stride=STRIDE), s_ray


def save(name, d):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def metrics_vert(s_ray, ray_count):
    cvs = np.zeros(ray_count - 2)
    for i in range(1, ray_count - 1):
        masked = np.ma.masked_equal(s_ray[:, 2, i], 0).compressed()
        cvs[i - 1] = 100 * np.std(masked) / np.mean(masked)
    return np.mean(cvs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    only = set(filter(None, ap.parse_args().only.split(",")))
    want = lambda k: not only or k in only  # noqa
    R = load_reference()
    os.makedirs(OUT, exist_ok=True)
    consts = dict(SIGMA=R.SIGMA, DELTA=R.DELTA, DELTA_S=R.DELTA_S, GOLD_TOL=R.GOLD_TOL, GOLD_RATIO=R.GOLD_RATIO,
                  N=R.N, DELTA_S_DIVISOR_FISHEYE=R.DELTA_S_DIVISOR_FISHEYE)
    if want("field"):
        gen_field(R)
    if want("step"):
        gen_step(R)
    if want("traj_interface"):
        # cfg1: interface, 16 rays, op6, default DELTA_S (preset rule :260 with ray_count=16)
        # (the preset keeps ray_count+1 angles and never uses the last one, Q9)
        th = np.linspace(2 * (np.pi / 60), np.pi / 2, 17)[:16]
        d, s_ray = run_traj(R, "interface", 6, R.DELTA_S, 91, rays=(th, np.ones(16) * -2))
        save("traj_interface_op6_16", d)
    if want("traj_interface_curv"):
        # the curvature advancement (op3/op4/op5) on the sigmoid interface, where its conditioning is worst (round 2)
        th = np.linspace(2 * (np.pi / 60), np.pi / 2, 17)[:16]
        for m in (3, 4, 5):
            d, s_ray = run_traj(R, "interface", m, R.DELTA_S, 91, rays=(th, np.ones(16) * -2))
            save(f"traj_interface_op{m}_16", d)
    if want("traj_interface_rest"):
        # the remaining isotropic methods on the same 16-ray interface fan (round 3): with these every scenario has every method
        th = np.linspace(2 * (np.pi / 60), np.pi / 2, 17)[:16]
        for m in (1, 2, 7, 8, 9):
            d, s_ray = run_traj(R, "interface", m, R.DELTA_S, 91, rays=(th, np.ones(16) * -2))
            save(f"traj_interface_op{m}_16", d)
    if want("traj_fisheye_fan"):
        # the 9-ray fisheye fan (both terminations) for every isotropic method (round 2)
        th = np.linspace(np.pi / 4, 3 * np.pi / 4, 9)
        for m in (1, 2, 3, 4, 5, 7, 8, 9):
            d, s_ray = run_traj(R, "fisheye", m, 2 * np.pi / 303, 304, rays=(th, np.array((1, 0))))
            save(f"traj_fisheye_op{m}_fan9", d)
    if want("traj_vert"):
        for m in range(1, 10):
            d, s_ray = run_traj(R, "vert_heterogeneous", m, R.DELTA_S, 91)
            d["cv_mean"] = metrics_vert(s_ray, 31)
            save(f"traj_vert_op{m}", d)
    if want("traj_aniso"):
        for m in (10, 11):
            d, s_ray = run_traj(R, "anisotropy", m, R.DELTA_S, 91)
            d["cv_mean"] = metrics_vert(s_ray, 31)
            save(f"traj_aniso_op{m}", d)
    if want("traj_fisheye"):
        for div, step in ((91, R.DELTA_S), (304, 2 * np.pi / 303)):
            d, s_ray = run_traj(R, "fisheye", 6, step, div)
            d["closure_pct"] = 100 * np.linalg.norm(np.array([1, 0]) - s_ray[-1, 0:2, 0]) / (2 * np.pi)
            save(f"traj_fisheye_op6_div{div}", d)
        # synthetic fan (SURVEY 8d cfg3 rule) at 9 rays: exercises termination in the fisheye box
        th = np.linspace(np.pi / 4, 3 * np.pi / 4, 9)
        d, s_ray = run_traj(R, "fisheye", 6, 2 * np.pi / 303, 304, rays=(th, np.array((1, 0))))
        save("traj_fisheye_op6_fan9", d)
    if want("sweep"):
        # DELTA_S calibration sweep (RT_bench.py:1296-1318): search_delta over every 10th candidate, op6
        for scen in ("interface", "fisheye", "vert_heterogeneous"):
            c, linx, liny, Z, z, grd = build_field(R, scen)
            R.op_interface, R.op_fish, R.op_vert_heterogeneous, R.op_anisotropy = c[9:13]   # globals search_delta reads (:953-957)
            if c[9]:
                divisors = np.arange(R.DELTA_S_DIVISOR_UPPER_LIMIT, R.DELTA_S_DIVISOR_LOWER_LIMIT - R.DELTA_STEP, -R.DELTA_STEP)
                opts = R.SIGMA / divisors
            elif c[10]:
                divisors = np.arange(R.DELTA_S_DIVISOR_FISHEYE_UPPER_LIMIT, R.DELTA_S_DIVISOR_FISHEYE_LOWER_LIMIT - R.DELTA_STEP_FISHEYE, -R.DELTA_STEP_FISHEYE)
                opts = 2 * np.pi / divisors
            else:
                divisors = np.arange(R.DELTA_S_DIVISOR_VERT_UPPER_LIMIT, R.DELTA_S_DIVISOR_VERT_LOWER_LIMIT - 2 * R.DELTA_STEP, -R.DELTA_STEP)
                opts = R.SIGMA / divisors
            sel = np.arange(0, len(divisors), 10)
            res = []
            t = time.time()
            for i in sel:
                r = R.search_delta(R.op6, z, grd, opts[i], divisors[i] + 1, SCEN[scen][0])
                if c[9]:
                    res.append([r[0], r[1]])
                elif c[10]:
                    res.append([r, 0.0])
                else:
                    cvs = np.zeros(c[1] - 2)
                    for k in range(1, c[1] - 1):
                        masked = np.ma.masked_equal(r[:, k], 0).compressed()
                        cvs[k - 1] = 100 * np.std(masked) / np.mean(masked)
                    res.append([np.mean(cvs), 0.0])
            print(f"sweep {scen}: {len(sel)} candidates {time.time() - t:.1f}s")
            save(f"sweep_{scen}_op6", dict(all_divisors=divisors, all_options=opts, sel=sel, results=np.array(res)))
    if want("isochrones"):
        # per-ray stage of the wavefront extraction (RT_bench.py:987-1003), with the reference's own calls
        from scipy.interpolate import PchipInterpolator
        for scen, m in (("vert_heterogeneous", 6), ("anisotropy", 11)):
            if scen == "anisotropy":
                continue   # the per-ray stage is the same code; the across-ray stage has its anisotropy fixture below
            d, s_ray = run_traj(R, scen, m, R.DELTA_S, 91)
            times = np.arange(0.05, 0.6, 0.05)
            out = np.full((len(times), 3, s_ray.shape[2]), np.nan)
            for it, travel_time in enumerate(times):
                for i in range(s_ray.shape[2]):
                    n = int(d["d_ray"][2, i]) + 1
                    t_ray = s_ray[:n, 4, i]
                    if np.max(t_ray) >= travel_time:
                        for q, col in enumerate((0, 1, 5)):
                            out[it, q, i] = PchipInterpolator(t_ray, s_ray[:n, col, i])(travel_time)
            save("isochrones_vert_op6", dict(times=times, points=out, theta=d["theta"], step=d["step"],
                                             max_size=d["max_size"], box=d["box"]))
    for key, scen, m, gam in (("wavefronts", "vert_heterogeneous", 6, 1), ("wavefronts_aniso", "anisotropy", 11, 3)):
        if not want(key):
            continue
        # across-ray stage of the wavefront extraction (RT_bench.py:1005-1026, 1043-1044), with the reference's own calls, on
        # the reference's own trajectories: vert op6 and -- where the ray angle is NOT the wavefront normal -- anisotropy
        # (gamma = 3) op11, 31 rays each (the latter takes ~90 s of reference time)
        from scipy.interpolate import PchipInterpolator
        if scen == "anisotropy":
            R.gamma = gam           # module-global read by op10/op11 (Q12); run_traj sets it too
        d, s_ray = run_traj(R, scen, m, R.DELTA_S, 91)
        times = np.arange(0.05, 0.6, 0.05)
        RC = s_ray.shape[2]
        res = dict(times=times, theta=d["theta"], step=d["step"], max_size=d["max_size"], box=d["box"], gamma=gam, method=m)
        for it, travel_time in enumerate(times):
            valid_ray_coord, angle_vector, rays = [], [], []
            for i in range(RC):
                n = int(d["d_ray"][2, i]) + 1
                t_ray = s_ray[:n, 4, i]
                if np.max(t_ray) >= travel_time:
                    x = PchipInterpolator(t_ray, s_ray[:n, 0, i])(travel_time)
                    y = PchipInterpolator(t_ray, s_ray[:n, 1, i])(travel_time)
                    angle_vector.append(PchipInterpolator(t_ray, s_ray[:n, 5, i])(travel_time))
                    valid_ray_coord.append([x, y]); rays.append(i)
            res[f"count{it}"] = len(valid_ray_coord)
            if len(valid_ray_coord) > 1:
                valid_ray_coord = np.array(valid_ray_coord); angle_vector = np.array(angle_vector)
                indices = np.argsort(valid_ray_coord[:, 1])
                ray_coord_sorted = valid_ray_coord[indices]
                pchip_interpolator = PchipInterpolator(ray_coord_sorted[:, 1], ray_coord_sorted[:, 0])
                dy_dx_original = pchip_interpolator.derivative()(ray_coord_sorted[:, 1])
                tangent_angles = np.pi / 2 - np.arctan(dy_dx_original)
                normal_angles = tangent_angles - np.pi / 2
                y_fine = np.linspace(min(ray_coord_sorted[:, 1]), max(ray_coord_sorted[:, 1]), 100)
                res.update({f"y{it}": ray_coord_sorted[:, 1], f"x{it}": ray_coord_sorted[:, 0], f"ray{it}": np.array(rays)[indices],
                            f"angle_rayorder{it}": angle_vector, f"dxdy{it}": dy_dx_original, f"normal{it}": normal_angles,
                            f"angle_diff_ref{it}": np.absolute(angle_vector - normal_angles),    # (:1032) as the reference zips it
                            f"x_fine{it}": pchip_interpolator(y_fine), f"y_fine{it}": y_fine})
        save("wavefronts_vert_op6" if scen != "anisotropy" else "wavefronts_aniso_op11", res)
    if want("consts"):
        save("constants", consts)


if __name__ == "__main__":
    main()
