"""CPU ORACLE (test infrastructure): NumPy right-hand sides and initial states
of the synthetic workloads named in BASELINE.json `configs` / SURVEY.md §8d.

They are the CPU twins of the built-in device RHS plugins
(extensisq_amd/csrc/rhs_builtin.hip); operation order is written out so that
the two agree to a few ulp.
"""
import numpy as np


# -- config 2 / 5: 2-D heat equation, 5-point Laplacian, Dirichlet 0 ----------
def heat2d_rhs(N):
    c = float((N + 1) ** 2)

    def fun(t, y):
        u = np.zeros((N + 2, N + 2))
        u[1:-1, 1:-1] = y.reshape(N, N)
        f = ((u[:-2, 1:-1] + u[2:, 1:-1]) + (u[1:-1, :-2] + u[1:-1, 2:])
             - 4.0 * u[1:-1, 1:-1])
        return (c * f).ravel()
    return fun


def heat2d_y0(N, seed=1234):
    x = np.arange(1, N + 1) / (N + 1)
    rng = np.random.default_rng(seed)
    u0 = np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    return (u0 + 0.1 * rng.standard_normal((N, N))).ravel()


def heat2d_rho(N):
    return 8.0 * (N + 1) ** 2


# -- config 3: 2-D Brusselator reaction-diffusion, periodic --------------------
BRUSS_A = 1.0
BRUSS_B = 3.4
BRUSS_ALPHA = 0.1


def bruss2d_rhs(N, alpha=BRUSS_ALPHA):
    d = alpha * float(N * N)

    def lap(w):
        return ((np.roll(w, 1, 0) + np.roll(w, -1, 0))
                + (np.roll(w, 1, 1) + np.roll(w, -1, 1)) - 4.0 * w)

    def fun(t, y):
        u = y[:N * N].reshape(N, N)
        v = y[N * N:].reshape(N, N)
        uuv = u * u * v
        du = (BRUSS_A + uuv - (BRUSS_B + 1.0) * u) + d * lap(u)
        dv = (BRUSS_B * u - uuv) + d * lap(v)
        return np.concatenate([du.ravel(), dv.ravel()])
    return fun


def bruss2d_y0(N):
    c = (np.arange(N) + 0.5) / N
    # first grid index = y direction (rows), second = x (columns)
    yy, xx = np.meshgrid(c, c, indexing="ij")
    u0 = 22.0 * yy * (1.0 - yy) ** 1.5
    v0 = 27.0 * xx * (1.0 - xx) ** 1.5
    return np.concatenate([u0.ravel(), v0.ravel()])


def bruss2d_rho(N, alpha=BRUSS_ALPHA):
    return 8.0 * alpha * N * N


# -- config 4: 3-D diffusion, 7-point Laplacian, Dirichlet 0 -------------------
def diff3d_rhs(N):
    c = float((N + 1) ** 2)

    def fun(t, y):
        u = np.zeros((N + 2, N + 2, N + 2))
        u[1:-1, 1:-1, 1:-1] = y.reshape(N, N, N)
        f = (((u[:-2, 1:-1, 1:-1] + u[2:, 1:-1, 1:-1])
              + (u[1:-1, :-2, 1:-1] + u[1:-1, 2:, 1:-1]))
             + (u[1:-1, 1:-1, :-2] + u[1:-1, 1:-1, 2:])
             - 6.0 * u[1:-1, 1:-1, 1:-1])
        return (c * f).ravel()
    return fun


def diff3d_y0(N):
    x = np.arange(1, N + 1) / (N + 1)
    s = np.sin(np.pi * x)
    return (s[:, None, None] * s[None, :, None] * s[None, None, :]).ravel()


def diff3d_rho(N):
    return 12.0 * (N + 1) ** 2


# -- small generic problems used by the golden fixtures -----------------------
def linear_rhs(lam):
    def fun(t, y):
        return lam * y
    return fun


def duffing_rhs(t, y):
    # forced Duffing oscillator of docs/Demo_BS5.ipynb (cell 1): y0 = [0, 0],
    # t in [0, 20]; published nfev: BS5 212, Ts5 341 (Demo_BS5.ipynb:137,175)
    return np.array([y[1], y[0] ** 3 / 6 - y[0] + 2 * np.sin(2.78535 * t)])


def rational_rhs(t, y):
    # scipy's classic test problem, tests/test_ivp.py:38-40; y0 = [1/3, 2/9]
    return np.array([y[1] / t,
                     y[1] * (y[0] + 2 * y[1] - 1) / (t * (y[0] - 1))])


def rational_sol(t):
    # tests/test_ivp.py:64-65
    return np.asarray((t / (t + 10), 10 * t / (t + 10) ** 2))


# -- SSV2stab known-answer problem: 3-D heat equation with a travelling tanh
#    front and time-dependent Dirichlet data (docs/Demo_SSV2stab.ipynb cells
#    "heat problem"; integer table at :350-356) ------------------------------
def tanh3d_problem(N=39):
    x = np.linspace(0.0, 1.0, N + 2)
    X, Y, Z = np.meshgrid(x, x, x)

    def exact(X, Y, Z, t):
        return np.tanh(5 * X + 10 * Y + 7.5 * Z - (2.5 + 5 * t))

    def source(t):
        s = exact(X, Y, Z, t)
        return 362.5 * (s - s ** 3) + 5 * s ** 2 - 5

    work = exact(X, Y, Z, 0.0)
    y0 = work[1:-1, 1:-1, 1:-1].copy().reshape(-1)
    inv_h2 = (N + 1.0) ** 2

    def fun(t, y):
        for ax in range(3):
            for side in (0, -1):
                idx = [slice(None)] * 3
                idx[ax] = side
                idx = tuple(idx)
                work[idx] = exact(X[idx], Y[idx], Z[idx], t)
        work[1:-1, 1:-1, 1:-1] = y.reshape(N, N, N)
        lap = inv_h2 * (-6 * work[1:-1, 1:-1, 1:-1]
                        + work[:-2, 1:-1, 1:-1] + work[2:, 1:-1, 1:-1]
                        + work[1:-1, :-2, 1:-1] + work[1:-1, 2:, 1:-1]
                        + work[1:-1, 1:-1, :-2] + work[1:-1, 1:-1, 2:])
        return (lap + source(t)[1:-1, 1:-1, 1:-1]).reshape(-1)

    def rho_jac(t, y):
        return 12.0 * inv_h2

    return fun, y0, rho_jac


def combustion3d_problem(N=40, lewis=0.9, alpha=1.0, delta=20.0, rate=5.0):
    """3-D combustion benchmark of the RKC paper as set up in the reference's
    docs/Demo_SSV2stab.ipynb (cells 1-3): concentration c and temperature T on
    an N^3 cell grid, mirror (Neumann) conditions on the three low faces, value 1
    (Dirichlet) on the three high faces, mesh width 1/(N + 1/2);
        c_t = lap c - D c exp(-delta/T),   L T_t = lap T + alpha D c exp(-delta/T)
    with D = R exp(delta) / (alpha delta).  State y = [c.ravel(), T.ravel()].
    Returns (fun, y0)."""
    damkohler = rate * np.exp(delta) / (alpha * delta)
    inv_h2 = (N + 0.5) ** 2
    n3 = N ** 3
    halo = [np.ones((N + 2,) * 3), np.ones((N + 2,) * 3)]   # high faces stay 1

    def laplacian(field, w):
        w[1:-1, 1:-1, 1:-1] = field
        w[0, :, :] = w[1, :, :]
        w[:, 0, :] = w[:, 1, :]
        w[:, :, 0] = w[:, :, 1]
        return inv_h2 * (-6 * w[1:-1, 1:-1, 1:-1]
                         + w[:-2, 1:-1, 1:-1] + w[2:, 1:-1, 1:-1]
                         + w[1:-1, :-2, 1:-1] + w[1:-1, 2:, 1:-1]
                         + w[1:-1, 1:-1, :-2] + w[1:-1, 1:-1, 2:])

    def fun(t, y):
        c = y[:n3].reshape(N, N, N)
        temp = y[n3:].reshape(N, N, N)
        lap_c = laplacian(c, halo[0])
        lap_t = laplacian(temp, halo[1])
        react = damkohler * c * np.exp(-delta / temp)
        return np.concatenate([(lap_c - react).reshape(-1),
                               ((lap_t + alpha * react) / lewis).reshape(-1)])

    return fun, np.ones(2 * n3)
