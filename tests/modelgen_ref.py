"""numpy float32 interpreter of a modelgen.NeuronModel (test infrastructure): the same nb_macro semantics as the
generated HIP (build_test/nb_macro/src/lib.rs:2259-2345), one rounding per operation, vectorised over neurons.
Used as the step function of numpy_ref.run_lattice."""
import numpy as np

import oracle_binding as ob

f32 = np.float32
_expf = np.vectorize(ob.expf, otypes=[np.float32])


def evaluate(e, env):
    kind = e[0]
    if kind == "num":
        return f32(e[1])
    if kind == "var":
        return env[e[1]]
    if kind == "neg":
        return (-evaluate(e[1], env)).astype(f32)
    if kind == "not":
        return ~evaluate(e[1], env)
    if kind == "call":
        return _expf(np.asarray(evaluate(e[2][0], env), f32))
    _, op, l, r = e
    a, b = evaluate(l, env), evaluate(r, env)
    if op in ("+", "-", "*", "/"):
        with np.errstate(all="ignore"):
            return {"+": np.add, "-": np.subtract, "*": np.multiply, "/": np.divide}[op](a, b, dtype=f32)
    if op in ("&&", "||"):
        return (a & b) if op == "&&" else (a | b)
    return {"==": np.equal, "!=": np.not_equal, ">=": np.greater_equal, "<=": np.less_equal, ">": np.greater,
            "<": np.less}[op](a, b)


def _run(stmts, env, mask=None):
    diffs = []
    for s in stmts:
        if s[0] == "diff":
            diffs.append((s[1], (evaluate(s[2], env) * env["dt"]).astype(f32)))
            continue
        _, name, op, expr = s
        val = evaluate(expr, env)
        if op != "=":
            val = evaluate(("bin", op[0], ("var", name), expr), env)
        new = np.broadcast_to(val, env[name].shape).astype(f32)
        env[name] = np.where(mask, new, env[name]).astype(f32) if mask is not None else new
    for name, d in diffs:
        env[name] = (env[name] + d).astype(f32)


def make_step(model):
    """step(state, i_in) -> spike mask; state: dict with current_voltage, dt, c_m, gap_conductance and the model's
    variables as float32 arrays (updated in place)."""
    def step(state, i_in):
        env = {"v": state["current_voltage"], "i": i_in, "dt": state["dt"], "c_m": state["c_m"],
               "gap_conductance": state["gap_conductance"]}
        for name, _ in model.variables:
            env[name] = state[name]
        _run(model.on_iteration, env)
        spike = np.broadcast_to(evaluate(model.spike_detection, env), env["v"].shape).copy()
        _run(model.on_spike, env, mask=spike)
        state["current_voltage"] = env["v"]
        for name, _ in model.variables:
            state[name] = env[name]
        return spike
    return step
