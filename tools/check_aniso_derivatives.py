#!/usr/bin/env python3
"""Closed-form derivatives of the anisotropic momentum curve M(t) = (cos t, gamma^2 sin t)/a(t), a = sqrt(gamma^2 sin^2 t + cos^2 t),
as rt_exact.h (ang_golden_aniso, `expand`) writes them, against mpmath's numerical derivatives at 40 digits."""
import mpmath as mp

mp.mp.dps = 40
worst = 0
for gam in (mp.mpf(3), mp.mpf("0.5"), mp.mpf("1.7")):
    G = gam ** 2 - 1
    g2 = gam ** 2
    af = lambda t: mp.sqrt(g2 * mp.sin(t) ** 2 + mp.cos(t) ** 2)
    Mx = lambda t: mp.cos(t) / af(t)
    My = lambda t: g2 * mp.sin(t) / af(t)
    for t in (mp.mpf("0.3"), mp.mpf("1.2"), mp.mpf("-2.0"), mp.mpf("1.5707"), mp.mpf("0.001")):
        s, c = mp.sin(t), mp.cos(t)
        a2 = af(t) ** 2
        A = 1 / af(t)
        d1 = (-g2 * s * A ** 3, g2 * c * A ** 3)
        d2 = (-g2 * c * A ** 5 * (1 - 2 * G * s * s), -g2 * s * A ** 5 * (1 + 3 * G - 2 * G * s * s))
        d3 = (g2 * s * A ** 7 * ((1 - 2 * G * s * s + 4 * G * c * c) * a2 + 5 * G * c * c * (1 - 2 * G * s * s)),
              -g2 * c * A ** 7 * ((1 + 3 * G - 6 * G * s * s) * a2 - 5 * G * s * s * (1 + 3 * G - 2 * G * s * s)))
        a1 = G * s * c * A
        a2d = G * A * ((c * c - s * s) - G * s * s * c * c * A ** 2)
        a3 = G * s * c * A * (-4 - 3 * G * (c * c - s * s) * A ** 2 + 3 * G * G * s * s * c * c * A ** 4)
        for k, d, av in ((1, d1, a1), (2, d2, a2d), (3, d3, a3)):
            worst = max(worst, abs(mp.diff(Mx, t, k) - d[0]), abs(mp.diff(My, t, k) - d[1]), abs(mp.diff(af, t, k) - av))
print("largest difference:", mp.nstr(worst, 3))
assert worst < mp.mpf("1e-30")
