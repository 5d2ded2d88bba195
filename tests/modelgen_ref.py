"""numpy float32 interpreter of a modelgen.NeuronModel (test infrastructure): the same nb_macro semantics as the
generated HIP (build_test/nb_macro/src/lib.rs:2259-2345), one rounding per operation, vectorised over neurons.
Used as the step function of numpy_ref.run_lattice."""
import numpy as np

import oracle_binding as ob

f32 = np.float32
_FUNCTIONS = {name: np.vectorize(fn, otypes=[np.float32]) for name, fn in
              (("exp", ob.expf), ("tanh", ob.tanhf), ("sinh", ob.sinhf), ("cosh", ob.coshf), ("sin", ob.sinf),
               ("cos", ob.cosf), ("tan", ob.tanf))}
_powif = np.vectorize(ob.powif, otypes=[np.float32])
_powf = np.vectorize(ob.powf, otypes=[np.float32])


def _rust_min(a, b):       # f32::min: a NaN operand yields the other one
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(a < b, a, b))).astype(f32)


def _rust_max(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.where(a > b, a, b))).astype(f32)


def evaluate(e, env):
    kind = e[0]
    if kind == "num":
        return f32(e[1])
    if kind == "bool":
        return np.bool_(e[1])
    if kind == "var":
        return (env[e[1]] != 0) if e[1] in env.get("$bools", ()) else env[e[1]]
    if kind == "neg":
        return (-evaluate(e[1], env)).astype(f32)
    if kind == "not":
        return ~evaluate(e[1], env)
    if kind == "call":
        args = [np.asarray(evaluate(a, env), f32) for a in e[2]]
        if e[1] == "min":
            return _rust_min(*args)
        if e[1] == "max":
            return _rust_max(*args)
        if e[1] == "isnan":
            return np.isnan(args[0])
        if e[1] == "powf":                         # `a ^ b`: (a.powf(b)), nb_macro lib.rs:135
            return _powf(*np.broadcast_arrays(*args))
        if e[1] == "rpow":                         # `a r^ b`: (a.max(0.0f32).powf(b)), lib.rs:136
            return _powf(*np.broadcast_arrays(_rust_max(args[0], f32(0.0)), args[1]))
        if e[1] == "heaviside":                    # nb_macro lib.rs:9176-9178: x < 0 -> 0, else x
            return np.where(args[0] < 0, f32(0), args[0]).astype(f32)
        return _FUNCTIONS[e[1]](args[0])
    if kind == "powi":
        return _powif(np.asarray(evaluate(e[1], env), f32), e[2])
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
        if s[0] == "if":
            shape = env["v"].shape
            remaining = np.ones(shape, bool) if mask is None else mask.copy()
            for cond, body in s[1]:
                take = remaining & np.broadcast_to(evaluate(cond, env), shape)
                _run(body, env, mask=take)
                remaining &= ~take
            if s[2] is not None:
                _run(s[2], env, mask=remaining)
            continue
        if s[0] == "scope":                       # inlined ion channel: own diffs, applied at the end of its body
            _run(s[1], env, mask=mask)
            continue
        if s[0] == "diff":
            diffs.append((s[1], (evaluate(s[2], env) * env["dt"]).astype(f32)))
            continue
        _, name, op, expr = s
        val = evaluate(expr, env)
        if op != "=":
            val = evaluate(("bin", op[0], ("var", name), expr), env)
        new = np.broadcast_to(val, env[name].shape).astype(f32)          # a bool becomes 1.0 / 0.0
        env[name] = np.where(mask, new, env[name]).astype(f32) if mask is not None else new
    for name, d in diffs:
        new = (env[name] + d).astype(f32)
        env[name] = np.where(mask, new, env[name]).astype(f32) if mask is not None else new


def make_spike_train_step(model):
    """iterate(state) -> is_spiking for a modelgen.SpikeTrainModel; state: current_voltage, is_spiking (float 0/1), dt,
    v_resting, v_th and the model's variables as float32 arrays (updated in place)"""
    def iterate(state):
        env = {"v": state["current_voltage"], "is_spiking": state["is_spiking"], "dt": state["dt"],
               "v_resting": state["v_resting"], "v_th": state["v_th"], "$bools": model.bools}
        for name, _ in model.variables:
            env[name] = state[name]
        _run(model.on_iteration, env)
        state["current_voltage"], state["is_spiking"] = env["v"], env["is_spiking"]
        for name, _ in model.variables:
            state[name] = env[name]
        return state["is_spiking"] != 0
    return iterate


def make_kinetics_step(model):
    """apply(state, **inputs): one apply_t_change (inputs is_spiking, v, dt) or apply_r_change (inputs t, dt) of a
    modelgen.KineticsModel; state holds the kinetics' own `t` / `r` and variables as float32 arrays"""
    def apply(state, **inputs):
        env = {"$bools": model.bools, model.state: state[model.state]}
        for k, val in inputs.items():
            env[k] = np.asarray(val) if k == "is_spiking" else np.asarray(val, f32)
        if "is_spiking" in env:
            env["is_spiking"] = env["is_spiking"].astype(f32)        # stored like every bool: 1.0 / 0.0
        env.setdefault("v", state[model.state])                      # _run sizes masks by env["v"]
        for name, _ in model.variables:
            env[name] = state[name]
        _run(model.on_iteration, env)
        state[model.state] = env[model.state]
        for name, _ in model.variables:
            state[name] = env[name]
    return apply


def refractoriness_effect(model, time_difference, v_th, v_resting, dt, decay=None, **variables):
    env = {"time_difference": f32(time_difference), "v_th": f32(v_th), "v_resting": f32(v_resting), "dt": f32(dt),
           "decay": f32(model.decay if decay is None else decay)}
    for name, default in model.variables:
        env[name] = f32(variables.get(name, default))
    return evaluate(model.effect, env)


def make_step(model):
    """step(state, i_in) -> spike mask; state: dict with current_voltage, dt, c_m, gap_conductance and the model's
    variables as float32 arrays (updated in place)."""
    def step(state, i_in):
        env = {"v": state["current_voltage"], "i": i_in, "dt": state["dt"], "c_m": state["c_m"],
               "gap_conductance": state["gap_conductance"], "$bools": model.bools}
        for name, _ in model.variables:
            env[name] = state[name]
        _run(model.on_iteration, env)
        spike = np.broadcast_to(evaluate(model.spike_detection, env), env["v"].shape).copy()
        _run(getattr(model, "after_detection", []), env)
        _run(model.on_spike, env, mask=spike)
        state["current_voltage"] = env["v"]
        for name, _ in model.variables:
            state[name] = env[name]
        return spike
    return step


# ---- stack program for the C oracle (oracle/snn_oracle.c::custom_run) -------------------------------------
_OPS = dict(END=0, CONST=1, LOAD=2, STORE=3, DIFF=4, NEG=5, NOT=6, ADD=7, SUB=8, MUL=9, DIV=10, EXP=11, EQ=12, NE=13,
            GE=14, LE=15, GT=16, LT=17, AND=18, OR=19, JZ=20, JMP=21, TANH=22, SINH=23, COSH=24, MIN=25, MAX=26,
            HEAVISIDE=27, POWI=28, MARK=29, FLUSH=30, RC_UPDATE=31, RC_SET=32, RC_GET=33, NT_APPLY=34, SIN=35, COS=36, TAN=37, ISNAN=38,
            POWF=39, RPOW=40)
_BIN = {"+": "ADD", "-": "SUB", "*": "MUL", "/": "DIV", "==": "EQ", "!=": "NE", ">=": "GE", "<=": "LE", ">": "GT",
        "<": "LT", "&&": "AND", "||": "OR"}
_BASE_SLOTS = {"v": 0, "i": 1, "dt": 2, "c_m": 3, "gap_conductance": 4}


_ST_SLOTS = {"v": 0, "is_spiking": 1, "dt": 2, "v_resting": 3, "v_th": 4}
_NT_SLOTS = {"t": 0, "is_spiking": 1, "dt": 2, "v": 3}
_RC_SLOTS = {"r": 0, "t": 1, "dt": 2}
_REFR_SLOTS = {"time_difference": 0, "v_th": 1, "dt": 2, "v_resting": 3, "decay": 4}


def compile_program(model, base=None, blocks=None):
    """(code int32[], consts float32[], section starts) of the oracle's stack machine.  Default: a neuron model's
    on_iteration / spike_detection / on_spike; `blocks` = [("statements", stmts) | ("expression", e)] otherwise."""
    slots = dict(_BASE_SLOTS if base is None else base)
    for k, (name, _) in enumerate(model.variables):
        slots[name] = 5 + k
    code, consts = [], []

    def emit_expr(e):
        kind = e[0]
        if kind in ("num", "bool"):
            consts.append(np.float32(e[1]))
            code.extend([_OPS["CONST"], len(consts) - 1])
        elif kind == "var":
            code.extend([_OPS["LOAD"], slots[e[1]]])
        elif kind in ("neg", "not"):
            emit_expr(e[1])
            code.append(_OPS["NEG" if kind == "neg" else "NOT"])
        elif kind == "call":
            for a in e[2]:
                emit_expr(a)
            code.append(_OPS[e[1].upper()])
        elif kind == "powi":
            emit_expr(e[1])
            code.extend([_OPS["POWI"], e[2]])
        elif kind == "rc_get":
            emit_expr(e[1])
            emit_expr(e[2])
            code.append(_OPS["RC_GET"])
        else:
            emit_expr(e[2])
            emit_expr(e[3])
            code.append(_OPS[_BIN[e[1]]])

    def emit_statements(stmts):
        for s in stmts:
            if s[0] == "if":
                exits = []
                for cond, body in s[1]:
                    emit_expr(cond)
                    code.extend([_OPS["JZ"], -1])
                    skip = len(code) - 1
                    emit_statements(body)
                    code.extend([_OPS["JMP"], -1])
                    exits.append(len(code) - 1)
                    code[skip] = len(code)
                if s[2] is not None:
                    emit_statements(s[2])
                for e in exits:
                    code[e] = len(code)
                continue
            if s[0] in ("rc_update", "nt_apply"):
                code.append(_OPS["RC_UPDATE" if s[0] == "rc_update" else "NT_APPLY"])
                continue
            if s[0] == "rc_set":
                emit_expr(s[1])
                code.append(_OPS["RC_SET"])
                continue
            if s[0] == "scope":
                code.append(_OPS["MARK"])
                emit_statements(s[1])
                code.append(_OPS["FLUSH"])
                continue
            if s[0] == "diff":
                emit_expr(s[2])
                code.extend([_OPS["DIFF"], slots[s[1]]])
            else:
                _, name, op, expr = s
                emit_expr(expr if op == "=" else ("bin", op[0], ("var", name), expr))
                code.extend([_OPS["STORE"], slots[name]])

    def emit_block(stmts):
        start = len(code)
        emit_statements(stmts)
        code.append(_OPS["END"])
        return start

    if blocks is None:
        blocks = [("statements", model.on_iteration), ("expression", model.spike_detection),
                  ("statements", model.on_spike)]
        if getattr(model, "on_electrochemical_iteration", None) is not None:
            blocks.append(("statements", model.on_electrochemical_iteration))
    starts = []
    for what, item in blocks:
        if what == "statements":
            starts.append(emit_block(item))
        else:
            starts.append(len(code))
            emit_expr(item)
            if item is getattr(model, "spike_detection", None):      # balanced statements: the flag stays on the stack
                emit_statements(getattr(model, "after_detection", []))
            code.append(_OPS["END"])
    assert _stack_depth(code) <= 64, "expression too deep for the oracle's 64-entry evaluation stack"
    return (np.array(code, np.int32), np.array(consts if consts else [0.0], np.float32), np.array(starts, np.uint32))


def _stack_depth(code):
    """upper bound of the evaluation-stack depth of a compiled program (straight-line scan; jumps keep the depth)"""
    push = {_OPS["CONST"], _OPS["LOAD"]}
    with_operand = {_OPS[k] for k in ("CONST", "LOAD", "STORE", "DIFF", "JZ", "JMP", "POWI")}
    pop1 = {_OPS[k] for k in ("STORE", "DIFF", "JZ", "RC_SET")}
    pop_binary = {_OPS[k] for k in ("ADD", "SUB", "MUL", "DIV", "EQ", "NE", "GE", "LE", "GT", "LT", "AND", "OR", "MIN",
                                    "MAX", "RC_GET", "POWF", "RPOW")}
    depth = worst = pc = 0
    while pc < len(code):
        op = code[pc]
        pc += 2 if op in with_operand else 1
        depth += 1 if op in push else -1 if (op in pop1 or op in pop_binary) else 0
        if op == _OPS["END"]:
            depth = 0
        worst = max(worst, depth)
    return worst


def attach_spike_train(net, model):
    """Make an oracle Net (created with st_kind=ob.ST_CUSTOM) iterate the generated spike train `model`."""
    code, consts, _ = compile_program(model, _ST_SLOTS, [("statements", model.on_iteration)])
    net.st_custom_model = model
    net.arr["st_custom_code"], net.arr["st_custom_consts"] = code, consts
    net.st_custom_nvars = len(model.variables)
    net.arr["st_custom_vars"] = np.zeros((max(1, len(model.variables)), net.n_cells), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["st_custom_vars"][k] = default
    net["st_current_voltage"] = model.mandatory["current_voltage"]
    net["st_dt"] = model.mandatory["dt"]
    net["st_v_resting"] = model.mandatory["v_resting"]
    net["st_v_th"] = model.mandatory["v_th"]
    return net


def attach_refractoriness(net, model):
    """Give every spike-train cell of an oracle Net the generated refractoriness `model` (kind 2)."""
    code, consts, _ = compile_program(model, _REFR_SLOTS, [("expression", model.effect)])
    net.refr_model = model
    net.arr["refr_code"], net.arr["refr_consts"] = code, consts
    net.refr_nvars = len(model.variables)
    net.arr["refr_vars"] = np.zeros((max(1, len(model.variables)), net.n_cells), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["refr_vars"][k] = default
    net["st_refractoriness"] = ob.REFRACTORINESS_CUSTOM
    net["st_k"] = model.decay
    return net


def attach(net, model):
    """Make an oracle Net (created with model=ob.CUSTOM) step `model`: program + variable arrays + DSL defaults."""
    code, consts, sections = compile_program(model)
    net.custom_model = model
    net.arr["custom_code"], net.arr["custom_consts"] = code, consts
    net.custom_section = sections[:3]
    net.custom_has_chem = int(len(sections) == 4)
    net.custom_chem_section = int(sections[3]) if len(sections) == 4 else 0
    net.custom_nvars = len(model.variables)
    net.arr["custom_vars"] = np.zeros((max(1, len(model.variables)), net.n_neurons), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["custom_vars"][k] = default
    net["current_voltage"] = model.mandatory["current_voltage"]
    net["dt"] = model.mandatory["dt"]
    net["c_m"] = model.mandatory["c_m"]
    net["gap_conductance"] = model.mandatory["gap_conductance"]
    return net


def attach_nt_kinetics(net, model):
    """Make an oracle Net (created with nt_kind=ob.NT_CUSTOM) use the generated neurotransmitter kinetics `model`."""
    code, consts, _ = compile_program(model, _NT_SLOTS, [("statements", model.on_iteration)])
    net.nt_model = model
    net.arr["nt_code"], net.arr["nt_consts"] = code, consts
    net.nt_nvars = len(model.variables)
    nv = max(1, len(model.variables))
    net.arr["nt_custom_vars"] = np.zeros((nv, net.n_neurons, 3), np.float32)
    net.arr["st_nt_custom_vars"] = np.zeros((nv, max(1, net.n_cells), 3), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["nt_custom_vars"][k] = default
        net.arr["st_nt_custom_vars"][k] = default
    return net


def attach_receptor_kinetics(net, model):
    """Make an oracle Net (created with rc_kind=ob.RC_CUSTOM) use the generated receptor kinetics `model`."""
    code, consts, _ = compile_program(model, _RC_SLOTS, [("statements", model.on_iteration)])
    net.rc_model = model
    net.arr["rc_code"], net.arr["rc_consts"] = code, consts
    net.rc_nvars = len(model.variables)
    net.arr["rc_custom_vars"] = np.zeros((max(1, len(model.variables)), net.n_neurons, 3), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["rc_custom_vars"][k] = default
    return net


_RX_SLOTS = {"v": 0, "r": 1, "dt": 2, "t": 3}


def attach_receptors(net, model):
    """Give the generated neurons of an oracle Net (model=ob.CUSTOM) the generated receptor set `model`."""
    nty = len(model.types)
    blocks = [("statements", stmts) for _, stmts, _ in model.types]
    if getattr(model, "multi", False):          # several states per type: one kinetics program per type after the iterates
        blocks += [("statements", code) for code in model.kinetics_code]
    code, consts, starts = compile_program(model, _RX_SLOTS, blocks)
    net.rx_model = model
    net.arr["rx_code"], net.arr["rx_consts"] = code, consts
    net.rx_ntypes, net.rx_nvars = nty, len(model.variables)
    net.rx_multi = int(getattr(model, "multi", False))
    if net.rx_multi:
        net.rx_kin_section = np.concatenate([starts[nty:], np.zeros(3 - nty, np.uint32)]).astype(np.uint32)
        starts = starts[:nty]
    net.rx_section = np.concatenate([starts, np.zeros(3 - len(starts), np.uint32)]).astype(np.uint32)
    net.rx_current_index = np.array([(-1 if t[2] is None else t[2]) for t in model.types] + [-1] * (3 - len(model.types)),
                                    np.int32)
    net.arr["rx_vars"] = np.zeros((max(1, len(model.variables)), net.n_neurons), np.float32)
    for k, (_, default) in enumerate(model.variables):
        net.arr["rx_vars"][k] = default
    return net
