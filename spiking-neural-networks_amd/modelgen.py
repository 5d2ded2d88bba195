"""Model description -> HIP: the reference's `neuron_builder!` DSL (/root/reference/build_test/nb_macro/src/lib.rs;
grammar in src/pest_ast) -- `[neuron]`, `[ion_channel]`, `[spike_train]`, `[neural_refractoriness]`,
`[neurotransmitter_kinetics]`, `[receptor_kinetics]` and `[receptors]` blocks.  `parse(text)` reads a neuron (with its
ion channels), `parse_description(text)` every block; `hip_source` emits the header a generated library is compiled
with (`_lib.build_custom`).  The simplest description is one `[neuron]` block:

    [neuron]
        type: BasicIntegrateAndFire
        vars: e = 0, v_reset = -75, v_th = -55
        on_spike:
            v = v_reset
        spike_detection: v >= v_th
        on_iteration:
            dv/dt = (v - e) + i
    [end]

Semantics are those of the code nb_macro generates (lib.rs:108-215, 659-700, 962-1002, 2196-2345; the hand expansion
the reference tests against is build_test/nb_macro/tests/lif_reference.rs):
  * mandatory variables current_voltage (`v`) = 0, dt = 0.1, c_m = 1, gap_conductance = 10 unless listed in `vars`;
    `i` is the input current;
  * on_iteration runs its statements in order; `dx/dt = expr` computes `dx = (expr) * dt` at that point and every
    `x += dx` is applied after the last statement, in statement order;
  * electrical step (lib.rs:2266-2272): on_iteration; is_spiking = spike_detection; on_spike if spiking.  With
    neurotransmission (lib.rs:2318-2328): receptor kinetics and currents at the old voltage first, on_iteration,
    `v -= receptor currents * (dt / c_m)`, the neuron's own neurotransmitter release, then the spike handling (the
    Ionotropic AMPA/NMDA/GABA receptors of the hot path stand where the generated Rust uses its receptor type unless
    the description brings a [receptors] block);
  * every binary operation is one float32 operation, evaluated left to right as written (no contraction).

  * `[if] cond [then] ... [elseif] cond [then] ... [else] ... [end]` (nestable, lib.rs:405-470) in on_iteration and
    on_spike; differential equations stay at the top level of on_iteration (a branch-local `dx` would be out of
    scope where the generated Rust applies it).

  * `[ion_channel]` blocks (lib.rs:3960-4075) and `ion_channels: name = Type, ...` in the neuron: a channel has
    `vars`, optional `gating_vars` (BasicGatingVariable: alpha, beta, state = 0; backend ion_channels/mod.rs:14-45)
    and an on_iteration that assigns `current`; `g.update(dt)`, `g.init_state()` are the struct calls of a gate.  The
    neuron calls `name.update_current(v)` -- `(v, dt)` when the channel has a differential equation or passes `dt`
    to a gate (lib.rs:3967-3997) -- and reads `name.current`, `name.var`, `name.gate.state`.  The generator inlines
    the channel body at the call (the channel's own `x += dx` at the end of ITS body) and stores its fields as the
    neuron variables `name$var`, `name$gate$alpha|beta|state`, `name$current` (the reference's attribute names);
  * functions exp, tanh, sinh, cosh, sin, cos, tan, min, max, heaviside, isnan (lib.rs:9139-9182; heaviside as
    written there: x < 0 -> 0, else x; isnan yields a bool) and `base ^ n` with an integer literal n (`powf`; binds tighter than * and /,
    left-associative: pest_ast/mod.rs:183-186; a leading unary minus of the base ends up OUTSIDE the power, see
    _power).

  * `on_electrochemical_iteration:` (lib.rs:2280-2316) replaces the default chemical step by the listed statements:
    `receptors.update_receptor_kinetics(t, dt)`, `receptors.set_receptor_currents(<voltage>, dt)`,
    `receptors.get_receptor_currents(<dt>, <c_m>)` (an expression: total current * (dt / c_m)),
    `synaptic_neurotransmitters.apply_t_changes()`, assignments and differential equations (applied after the last
    statement); spike handling follows as in the electrical step;
  * bool variables (`flag = false`, tests/bool_vars.rs): stored as 1.0 / 0.0 in a float plane (set them to 0 or 1
    only), typed like the generated Rust -- conditions, `!`, `&&`, `||` take bools, arithmetic and comparisons take
    numbers, and a description rustc would refuse for mixing them is refused here.

  * `[spike_train]` blocks (lib.rs:4812-4905; `parse_description`): `vars` plus the mandatory dt = 0.1, v_resting = 0,
    v_th = 30, current_voltage = 0, is_spiking = false; iterate() = on_iteration (which writes `current_voltage` / `v`
    and the bool `is_spiking`), then the neurotransmitter update on the new flag -> spike-train model SNN_ST_CUSTOM;
  * `[neural_refractoriness]` blocks (lib.rs:5677-5762): `effect:` is one expression over v_th, v_resting, dt,
    time_difference = (timestep - last_firing_time) as f32, `decay` (always there, default 10000, stored where the
    built-in kinds keep k) and the block's own vars -> neural_refractoriness$kind 2.
  * `[neurotransmitter_kinetics]` (lib.rs:6468-6540; NT kinetics selector 100) and `[receptor_kinetics]`
    (lib.rs:6757-6826; receptor kinetics selector 100) blocks: on_iteration over the state (`t` / `r`, default 0), the
    block's vars -- one value per neurotransmitter type, attributes neurotransmitters$<name> /
    receptors$<TYPE>$r$kinetics$<name> -- and the inputs is_spiking, v, dt / t, dt.
  * `[receptors]` blocks (lib.rs:7017-7600) used by the neuron through `receptors: <type>`: optional top-level vars
    and one to three `neurotransmitter: <Name>` groups (they take the exchange's three slots in order), each with vars
    and an on_iteration over v, its receptor state `r`, its vars and the top-level vars.  set_receptor_currents runs
    the on_iteration of every receptor present (receptors$flags) in declaration order, get_receptor_currents sums the
    `current` variables; attributes receptors$<var>, receptors$<Name>$<var>, receptors$<Name>$r$kinetics$r.
    One library carries at most one neuron (with its receptor set), spike train, refractoriness and kinetics block
    of each kind.

`hip_source(model)` emits the header that csrc/snn_custom_model.hpp includes when the library is compiled with
-DSNN_CUSTOM_MODEL_HEADER; `_lib.build_custom(model)` compiles such a library.  `base ^ y` with any exponent is
libm's powf (an integer literal y keeps the folded forms), `base r^ y` is `base.max(0).powf(y)` (lib.rs:135-136).
Several receptor states per neurotransmitter (`receptors: ampa_r, nmda_r`, with the block's `kinetics:` type): every
state's r and kinetics variables become variables of the set (receptors$<Type>$<state>$kinetics$<var>) and the
kinetics run on each state of a type whose transmitter arrives -- the description of the reference's own Python
module (interface_gpu/lixirnet/src/lib.rs:22-79, examples_dsl.LIXIRNET) is built this way.  `spike_detection: continuous()`: the Rust the reference prints for it (lib.rs:984-990) reads
a `last_voltage` it never defines; the detector it spells out is the built-in HodgkinHuxleyNeuron's
(hodgkin_huxley/mod.rs:207-220: a peak above v_th), and that is what is generated, with last_voltage = the voltage at
the start of the iteration and the bool `was_increasing` as an extra variable.
"""
import re
import struct

MANDATORY = {"current_voltage": 0.0, "dt": 0.1, "c_m": 1.0, "gap_conductance": 10.0}
MAX_VARS = 32
FUNCTIONS = {"exp": 1, "tanh": 1, "sinh": 1, "cosh": 1, "sin": 1, "cos": 1, "tan": 1, "heaviside": 1, "isnan": 1,
             "min": 2, "max": 2}
MAX_POWER = 16
MAX_ST_VARS = 16
MAX_REFRACTORINESS_VARS = 8
MAX_KINETICS_VARS = 8
MAX_RECEPTOR_VARS = 32


class ModelError(ValueError):
    pass


# ---- tokens -----------------------------------------------------------------------------------------
_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*(?:\.[A-Za-z_][A-Za-z_0-9]*)*)|"
                    r"(\|\||&&|==|!=|>=|<=|[-+*/()<>!,^]))")


_RPOWER = re.compile(r"\s*r\^")


def _tokens(text):
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        # `r^` (pest_ast/mod.rs:46) is an operator only where an operator may stand: after an operand
        if out and (out[-1][0] in ("num", "name") or out[-1] == ("op", ")")):
            m = _RPOWER.match(text, pos)
            if m:
                out.append(("op", "r^"))
                pos = m.end()
                continue
        m = _TOKEN.match(text, pos)
        if not m:
            raise ModelError(f"cannot read expression at: {text[pos:]!r}")
        num, name, op = m.groups()
        out.append(("num", num) if num else ("name", name) if name else ("op", op))
        pos = m.end()
    return out


# ---- expressions: ("num", f) | ("var", name) | ("neg", e) | ("not", e) | ("bin", op, l, r) | ("call", name, [args])
#                   | ("powi", base, n)
_LEVELS = [("||",), ("&&",), ("==", "!=", ">=", "<=", ">", "<"), ("+", "-"), ("*", "/"), ("^", "r^")]


def _integer_literal(e):
    if e[0] == "neg":
        n = _integer_literal(e[1])
        return None if n is None else -n
    if e[0] == "num" and float(e[1]).is_integer() and abs(e[1]) <= MAX_POWER:
        return int(e[1])
    return None


def _power(base, n):
    """`base ^ n` as the generated Rust evaluates it: the reference prints a unary minus as "-<operand>" and the power
    as "(<lhs>.powf(n))" (lib.rs:126, 135), and in Rust the method call binds tighter than the minus -- `-s ^ 2` is
    -(s^2) (tests/timestep_dependent_ion_channel.rs compares it with `-self.s.state.powf(2.)`)."""
    if base[0] == "neg":
        return ("neg", _power(base[1], n))
    return ("powi", base, n)


def _general_power(base, exponent, restricted):
    """`base ^ y` with any exponent: (base.powf(y)); `base r^ y`: (base.max(0.0f32).powf(y)) (lib.rs:135-136) -- libm's
    powf on both sides, so powf(0, 0) = 1 as the OpenCL form spells out (lib.rs:347).  The unary minus of the base ends
    up outside, as in _power."""
    if base[0] == "neg":
        return ("neg", _general_power(base[1], exponent, restricted))
    return ("call", "rpow" if restricted else "powf", [base, exponent])


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def expr(self, level=0):
        if level == len(_LEVELS):
            return self.unary()
        lhs = self.expr(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in _LEVELS[level]:
            op = self.take()[1]
            rhs = self.expr(level + 1)
            if op == "^":
                n = _integer_literal(rhs)
                lhs = _power(lhs, n) if n is not None else _general_power(lhs, rhs, False)
            elif op == "r^":
                lhs = _general_power(lhs, rhs, True)
            else:
                lhs = ("bin", op, lhs, rhs)
        return lhs

    def unary(self):
        kind, val = self.peek()
        if kind == "op" and val == "-":
            self.take()
            return ("neg", self.unary())
        if kind == "op" and val == "!":
            self.take()
            return ("not", self.unary())
        return self.atom()

    def atom(self):
        kind, val = self.take()
        if kind == "num":
            return ("num", float(val))
        if kind == "name":
            if self.peek() == ("op", "("):
                self.take()
                args = []
                if self.peek() != ("op", ")"):
                    args.append(self.expr())
                    while self.peek() == ("op", ","):
                        self.take()
                        args.append(self.expr())
                if self.take() != ("op", ")"):
                    raise ModelError("missing ')' after function arguments")
                if val == "receptors.get_receptor_currents":
                    if len(args) != 2:
                        raise ModelError("receptors.get_receptor_currents takes (dt, c_m)")
                    return ("rc_get", args[0], args[1])
                if val not in FUNCTIONS:
                    raise ModelError(f"function {val}() is not supported ({', '.join(sorted(FUNCTIONS))})")
                if len(args) != FUNCTIONS[val]:
                    raise ModelError(f"{val}() takes {FUNCTIONS[val]} argument(s)")
                return ("call", val, args)
            if val in ("true", "false"):
                return ("bool", val == "true")
            return ("var", val)
        if (kind, val) == ("op", "("):
            e = self.expr()
            if self.take() != ("op", ")"):
                raise ModelError("missing ')'")
            return e
        raise ModelError(f"unexpected token {val!r}")


def parse_expr(text):
    p = _Parser(_tokens(text))
    e = p.expr()
    if p.peek() != (None, None):
        raise ModelError(f"trailing input in expression {text!r}")
    return e


# ---- statements: ("diff", name, expr) | ("assign", name, op, expr) | ("struct_call", name, method, [args])
#                  | ("if", ...) | ("scope", stmts): an inlined channel body, its diffs applied at its end
_NAME = r"[A-Za-z_][A-Za-z_0-9]*"
_DIFF = re.compile(rf"^d({_NAME})\s*/\s*dt\s*=\s*(.+)$")
_ASSIGN = re.compile(rf"^({_NAME}(?:\.{_NAME})*)\s*(=|\+=|-=|\*=|/=)\s*(.+)$")
_STRUCT_CALL = re.compile(rf"^({_NAME}(?:\.{_NAME})*)\.({_NAME})\s*\((.*)\)$")


def _statement(line):
    m = _DIFF.match(line)
    if m:
        return ("diff", m.group(1), parse_expr(m.group(2)))
    m = _ASSIGN.match(line)
    if m and not m.group(3).startswith("="):
        return ("assign", m.group(1), m.group(2), parse_expr(m.group(3)))
    m = _STRUCT_CALL.match(line)
    if m:
        inner = m.group(3).strip()
        if not inner:
            return ("struct_call", m.group(1), m.group(2), [])
        # arguments: parse `f(a, b)` and take the argument list
        p = _Parser(_tokens(inner))
        args = [p.expr()]
        while p.peek() == ("op", ","):
            p.take()
            args.append(p.expr())
        if p.peek() != (None, None):
            raise ModelError(f"cannot read the arguments of {line!r}")
        return ("struct_call", m.group(1), m.group(2), args)
    raise ModelError(f"cannot read statement {line!r}")


_IF = re.compile(r"^\[(if|elseif)\]\s*(.+?)\s*\[then\]$")


def _block(lines, pos=0, nested=False):
    """statements of lines[pos:] up to the [elseif] / [else] / [end] that closes the enclosing [if];
    ("if", [(cond, stmts), ...], else_stmts | None)"""
    out = []
    while pos < len(lines):
        line = lines[pos]
        m = _IF.match(line)
        if m and m.group(1) == "if":
            branches, otherwise = [], None
            cond = parse_expr(m.group(2))
            body, pos = _block(lines, pos + 1, nested=True)
            branches.append((cond, body))
            while True:
                if pos >= len(lines):
                    raise ModelError("[if] without [end]")
                m2 = _IF.match(lines[pos])
                if m2 and m2.group(1) == "elseif":
                    body, pos2 = _block(lines, pos + 1, nested=True)
                    branches.append((parse_expr(m2.group(2)), body))
                    pos = pos2
                elif lines[pos] == "[else]":
                    otherwise, pos = _block(lines, pos + 1, nested=True)
                elif lines[pos] == "[end]":
                    pos += 1
                    break
                else:
                    raise ModelError(f"expected [elseif] / [else] / [end], got {lines[pos]!r}")
            out.append(("if", branches, otherwise))
            continue
        if line in ("[else]", "[end]") or (m and m.group(1) == "elseif"):
            if not nested:
                raise ModelError(f"{line!r} without [if]")
            return out, pos
        out.append(_statement(line))
        pos += 1
    if nested:
        raise ModelError("[if] without [end]")
    return out, pos


class NeuronModel:
    def __init__(self, name, variables, on_iteration, spike_detection, on_spike):
        self.name = name
        self.variables = variables          # [(name, default)] in declaration order, without the mandatory ones;
                                            # the fields of ion channel `c` follow as c$var, c$gate$alpha, ..., c$current
        self.mandatory = dict(MANDATORY)    # defaults of current_voltage / dt / c_m / gap_conductance
        self.on_iteration, self.spike_detection, self.on_spike = on_iteration, spike_detection, on_spike
        self.ion_channels = []              # [(instance name, channel type name)]
        self.on_electrochemical_iteration = None   # statements replacing the default chemical step (lib.rs:2280-2316)
        self.receptors = None               # name of the [receptors] block the neuron uses (default: ionotropic set)
        self.after_detection = []           # statements run right after spike_detection is evaluated (continuous())
        self.bools = set()                  # variables declared true / false: stored as 1.0 / 0.0


class IonChannel:
    def __init__(self, name, variables, gating_vars, on_iteration, bools=()):
        self.name, self.variables, self.gating_vars, self.on_iteration = name, variables, gating_vars, on_iteration
        self.bools = set(bools)
        # lib.rs:3967-3997: update_current takes dt when the body has a differential equation or hands `dt` to a call
        self.uses_timestep = any(
            st[0] == "diff" or (st[0] == "struct_call" and any(a == ("var", "dt") for a in st[3]))
            for st in on_iteration)

    def fields(self):
        """[(field, default)] in the order of the generated struct (lib.rs:4006-4035): vars, gates, current"""
        out = list(self.variables)
        for g in self.gating_vars:
            out += [(f"{g}$alpha", 0.0), (f"{g}$beta", 0.0), (f"{g}$state", 0.0)]
        if "current" not in dict(self.variables):
            out.append(("current", 0.0))
        return out


_BLOCK_HEAD = re.compile(r"\[(neuron|ion_channel|receptors|neurotransmitter_kinetics|receptor_kinetics|spike_train|"
                         r"neural_refractoriness)\]")


def _split_blocks(text):
    """[(kind, body lines)] of the top-level [kind] ... [end] blocks"""
    lines = [l.strip() for l in text.splitlines() if l.strip()]
    blocks, k = [], 0
    while k < len(lines):
        m = _BLOCK_HEAD.fullmatch(lines[k])
        if not m:
            raise ModelError(f"text outside a block: {lines[k]!r}")
        depth, start, k = 0, k + 1, k + 1
        while True:
            if k >= len(lines):
                raise ModelError(f"[{m.group(1)}] without [end]")
            l = lines[k]
            if _BLOCK_HEAD.fullmatch(l):
                raise ModelError(f"[{m.group(1)}] without [end]")
            if _IF.match(l) and l.startswith("[if]"):
                depth += 1
            elif l == "[end]":
                if depth == 0:
                    break
                depth -= 1
            k += 1
        blocks.append((m.group(1), lines[start:k]))
        k += 1
    return blocks


def _sections(body, allowed):
    sections, current = {}, None
    head = re.compile(r"^(type|vars|gating_vars|on_spike|spike_detection|on_iteration|on_electrochemical_iteration|"
                      r"ion_channels|kinetics|receptors)\s*:\s*(.*)$")
    for line in body:
        m = head.match(line)
        if m:
            current = m.group(1)
            sections.setdefault(current, [])
            if m.group(2):
                sections[current].append(m.group(2))
        elif current is None:
            raise ModelError(f"text outside a section: {line!r}")
        else:
            sections[current].append(line)
    for name in sections:
        if name not in allowed:
            raise ModelError(f"section '{name}' is not supported")
    name = (sections.get("type") or [""])[0].strip()
    if not re.fullmatch(_NAME, name):
        raise ModelError(f"bad or missing type name {name!r}")
    return sections, name


def _variables(items, reserved, bools):
    """[(name, default)]; bool variables (stored as 1.0 / 0.0) are added to the set `bools`"""
    out = []
    for item in ",".join(items).split(","):
        item = item.strip()
        if not item:
            continue
        m = re.fullmatch(rf"({_NAME})\s*=\s*(-?\s*[0-9.eE+-]+|true|false)", item)
        if not m:
            raise ModelError(f"cannot read variable {item!r}")
        var = m.group(1)
        if m.group(2) in ("true", "false"):
            value = 1.0 if m.group(2) == "true" else 0.0
            bools.add(var)
        else:
            value = float(m.group(2).replace(" ", ""))
        if var in reserved:
            raise ModelError(f"'{var}' is reserved")
        if var in dict(out):
            raise ModelError(f"variable {var} is defined twice")
        out.append((var, value))
    return out


def _walk(stmts):
    for s in stmts:
        yield s
        if s[0] == "if":
            for _, body in s[1]:
                yield from _walk(body)
            if s[2] is not None:
                yield from _walk(s[2])
        elif s[0] == "scope":
            yield from _walk(s[1])


def _map_expr(e, rename):
    kind = e[0]
    if kind in ("num", "bool"):
        return e
    if kind == "var":
        return rename(e[1])
    if kind in ("neg", "not"):
        return (kind, _map_expr(e[1], rename))
    if kind == "call":
        return ("call", e[1], [_map_expr(a, rename) for a in e[2]])
    if kind == "powi":
        return ("powi", _map_expr(e[1], rename), e[2])
    if kind == "rc_get":
        return ("rc_get", _map_expr(e[1], rename), _map_expr(e[2], rename))
    return ("bin", e[1], _map_expr(e[2], rename), _map_expr(e[3], rename))


def _expr_vars(e):
    if e[0] == "var":
        yield e[1]
    for sub in e[1:]:
        if isinstance(sub, tuple):
            yield from _expr_vars(sub)
        elif isinstance(sub, list):
            for x in sub:
                yield from _expr_vars(x)


def _parse_channel(body):
    sections, name = _sections(body, ("type", "vars", "gating_vars", "on_iteration"))
    if not sections.get("on_iteration"):
        raise ModelError(f"ion channel {name}: section 'on_iteration' is missing")
    gates = [g.strip() for g in ",".join(sections.get("gating_vars", [])).split(",") if g.strip()]
    for g in gates:
        if not re.fullmatch(_NAME, g):
            raise ModelError(f"bad gating variable name {g!r}")
    bools = set()
    variables = _variables(sections.get("vars", []), ("v", "i", "dt") + tuple(gates), bools)
    stmts = _block(sections["on_iteration"])[0]
    for st in stmts:
        if st[0] == "if":
            for inner in _walk([st]):
                if inner[0] == "diff":
                    raise ModelError("differential equations belong to the top level of on_iteration")
    if "current" in bools:
        raise ModelError(f"ion channel {name}: 'current' is a number")
    return IonChannel(name, variables, gates, stmts, bools)


def _inline_channel(inst, ch, args):
    """the body of `inst.update_current(args)` over the neuron's variables inst$field"""
    if len(args) != (2 if ch.uses_timestep else 1):
        raise ModelError(f"{inst}.update_current takes {'(v, dt)' if ch.uses_timestep else '(v)'} for channel {ch.name}")
    fields = dict(ch.fields())
    for a in args:
        for n in _expr_vars(a):
            if n.startswith(inst + "$"):
                raise ModelError(f"the arguments of {inst}.update_current must not read {inst}'s own fields")

    def rename(n):
        if n == "v":
            return args[0]
        if n == "dt":
            if "dt" in fields:
                return ("var", f"{inst}$dt")
            if not ch.uses_timestep:
                raise ModelError(f"ion channel {ch.name} reads dt but its update_current takes no timestep "
                                 "(no differential equation, no dt argument to a gate)")
            return args[1]
        key = n.replace(".", "$")
        if key not in fields:
            raise ModelError(f"ion channel {ch.name}: unknown variable {n!r}")
        return ("var", f"{inst}${key}")

    def target(n):
        key = n.replace(".", "$")
        if key not in fields:
            raise ModelError(f"ion channel {ch.name}: cannot assign to {n!r}")
        return f"{inst}${key}"

    def convert(stmts):
        out = []
        for st in stmts:
            if st[0] == "if":
                out.append(("if", [(_map_expr(c, rename), convert(b)) for c, b in st[1]],
                            None if st[2] is None else convert(st[2])))
            elif st[0] == "diff":
                if args[1] != ("var", "dt"):
                    raise ModelError(f"{inst}.update_current: a channel with a differential equation takes the "
                                     "neuron's dt as its timestep")
                out.append(("diff", target(st[1]), _map_expr(st[2], rename)))
            elif st[0] == "assign":
                out.append(("assign", target(st[1]), st[2], _map_expr(st[3], rename)))
            else:                                  # gate.update(dt) / gate.init_state()
                _, gate, method, cargs = st
                if gate not in ch.gating_vars:
                    raise ModelError(f"ion channel {ch.name}: {gate!r} is not a gating variable")
                al, be, stt = (("var", f"{inst}${gate}${f}") for f in ("alpha", "beta", "state"))
                if method == "update" and len(cargs) == 1:
                    # BasicGatingVariable::update (ion_channels/mod.rs:40-44)
                    step = _map_expr(cargs[0], rename)
                    alpha_state = ("bin", "*", al, ("bin", "-", ("num", 1.0), stt))
                    beta_state = ("bin", "*", be, stt)
                    out.append(("assign", stt[1], "+=", ("bin", "*", step, ("bin", "-", alpha_state, beta_state))))
                elif method == "init_state" and not cargs:          # :35-37
                    out.append(("assign", stt[1], "=", ("bin", "/", al, ("bin", "+", al, be))))
                else:
                    raise ModelError(f"gating variables have update(dt) and init_state(), not {method}")
        return out

    return ("scope", convert(ch.on_iteration))


class SpikeTrainModel:
    """[spike_train] (lib.rs:4812-4905): iterate() = on_iteration, then the neurotransmitter update, returns
    is_spiking.  `v` / `current_voltage` and the bool `is_spiking` are written by the description."""
    MANDATORY = {"dt": 0.1, "v_resting": 0.0, "v_th": 30.0, "current_voltage": 0.0}

    def __init__(self, name, variables, bools, on_iteration):
        self.name, self.variables, self.on_iteration = name, variables, on_iteration
        self.bools = set(bools) | {"is_spiking"}
        self.mandatory = dict(self.MANDATORY)


class RefractorinessModel:
    """[neural_refractoriness] (lib.rs:5677-5762): get_effect(timestep, last_firing_time, v_th, v_resting, dt) with
    time_difference = (timestep - last_firing_time) as f32; `decay` (default 10000) always exists."""
    def __init__(self, name, variables, effect):
        self.name, self.variables, self.effect = name, variables, effect       # variables without `decay`
        self.decay = 10000.0


class KineticsModel:
    """[neurotransmitter_kinetics] (lib.rs:6468-6540): apply_t_change(neuron) = on_iteration over the state `t`, the
    block's vars and the releasing cell's is_spiking / v / dt.  [receptor_kinetics] (lib.rs:6757-6826):
    apply_r_change(t, dt) = on_iteration over the state `r`, the vars, the neurotransmitter concentration `t` and dt.
    `state` is "t" or "r" (default 0 unless listed in vars); every variable exists once per neurotransmitter type."""
    def __init__(self, name, state, variables, bools, on_iteration, state_default):
        self.name, self.state, self.variables, self.on_iteration = name, state, variables, on_iteration
        self.bools, self.state_default = set(bools), state_default


class ReceptorsModel:
    """[receptors] (lib.rs:7017-7600): up to three neurotransmitter types, each with its own vars and an on_iteration
    over v, its receptor state `r`, its vars and the block's top-level vars; set_receptor_currents runs the
    on_iterations of the receptors present in declaration order, get_receptor_currents sums the `current` variables.
    `variables` is the flattened table: top-level vars under their names, the vars of type T as "T$name"."""
    def __init__(self, name, types, variables, bools):
        self.name, self.types, self.variables, self.bools = name, types, variables, set(bools)
        # types: [(type name, statements, index of its `current` in `variables` or None)]


class Description:
    def __init__(self, neuron=None, spike_train=None, refractoriness=None, nt_kinetics=None, receptor_kinetics=None,
                 receptors=None):
        self.neuron, self.spike_train, self.refractoriness = neuron, spike_train, refractoriness
        self.nt_kinetics, self.receptor_kinetics, self.receptors = nt_kinetics, receptor_kinetics, receptors

    def parts(self):
        return [m for m in (self.neuron, self.spike_train, self.refractoriness, self.nt_kinetics,
                            self.receptor_kinetics, self.receptors) if m is not None]

    @property
    def name(self):
        return "_".join(m.name for m in self.parts())


def _convert_plain(stmts, rename, assignable, where):
    out = []
    for st in stmts:
        if st[0] == "if":
            out.append(("if", [(_map_expr(c, rename), _convert_plain(b, rename, assignable, "an [if] branch"))
                               for c, b in st[1]],
                        None if st[2] is None else _convert_plain(st[2], rename, assignable, "an [if] branch")))
        elif st[0] == "struct_call":
            raise ModelError(f"cannot call {st[1]}.{st[2]}() here")
        else:
            if st[0] == "diff" and where != "on_iteration":
                raise ModelError(f"differential equations belong to the top level of on_iteration, not {where}")
            tgt = rename(st[1])[1]                     # the target goes through the same name table as the reads
            if tgt not in assignable:
                raise ModelError(f"cannot assign to {st[1]!r}")
            out.append((st[0], tgt) + tuple(st[2:-1]) + (_map_expr(st[-1], rename),))
    return out


def _parse_spike_train(body):
    sections, name = _sections(body, ("type", "vars", "on_iteration"))
    if not sections.get("on_iteration"):
        raise ModelError(f"spike train {name}: section 'on_iteration' is missing")
    bools = set()
    variables = []
    model = SpikeTrainModel(name, variables, bools, [])
    for var, value in _variables(sections.get("vars", []), ("v", "i", "last_firing_time"), bools):
        if var == "is_spiking":
            if var not in bools:
                raise ModelError("'is_spiking' is a bool")
            continue                                   # always starts false on the device (spike_train/mod.rs defaults)
        if var in model.mandatory:
            if var in bools:
                raise ModelError(f"'{var}' is a number")
            model.mandatory[var] = value
        else:
            variables.append((var, value))
    if len(variables) > MAX_ST_VARS:
        raise ModelError(f"more than {MAX_ST_VARS} spike-train variables")
    model.bools = set(bools) | {"is_spiking"}
    known = {"v", "is_spiking", "dt", "v_resting", "v_th"} | {n for n, _ in variables}

    def rename(n):
        n = "v" if n == "current_voltage" else n
        if n not in known:
            raise ModelError(f"unknown variable {n!r}")
        return ("var", n)

    model.on_iteration = _convert_plain(_block(sections["on_iteration"])[0], rename,
                                        known - {"dt", "v_resting", "v_th"}, "on_iteration")
    _check_types(model.on_iteration, model.bools)
    return model


def _parse_refractoriness(body):
    sections, current = {}, None
    for line in body:
        m = re.match(r"^(type|vars|effect)\s*:\s*(.*)$", line)
        if m:
            current = m.group(1)
            sections.setdefault(current, [])
            if m.group(2):
                sections[current].append(m.group(2))
        elif current is None:
            raise ModelError(f"text outside a section: {line!r}")
        else:
            sections[current].append(line)
    name = (sections.get("type") or [""])[0].strip()
    if not re.fullmatch(_NAME, name):
        raise ModelError(f"bad or missing type name {name!r}")
    if not sections.get("effect"):
        raise ModelError(f"refractoriness {name}: section 'effect' is missing")
    bools = set()
    variables = _variables(sections.get("vars", []), ("v_th", "v_resting", "dt", "time_difference", "last_firing_time"),
                           bools)
    if bools:
        raise ModelError("bool variables have no use in a refractoriness effect")
    model = RefractorinessModel(name, [(n, d) for n, d in variables if n != "decay"], None)
    model.decay = dict(variables).get("decay", 10000.0)          # lib.rs:5685-5706
    if len(model.variables) > MAX_REFRACTORINESS_VARS:
        raise ModelError(f"more than {MAX_REFRACTORINESS_VARS} refractoriness variables")
    known = {"v_th", "v_resting", "dt", "time_difference", "decay"} | {n for n, _ in model.variables}

    def rename(n):
        if n not in known:
            raise ModelError(f"unknown variable {n!r}")
        return ("var", n)

    model.effect = _map_expr(parse_expr(" ".join(sections["effect"])), rename)
    _check_types([], set(), number=model.effect)
    return model


def _parse_kinetics(body, state):
    what = "neurotransmitter kinetics" if state == "t" else "receptor kinetics"
    sections, name = _sections(body, ("type", "vars", "on_iteration"))
    if not sections.get("on_iteration"):
        raise ModelError(f"{what} {name}: section 'on_iteration' is missing")
    bools = set()
    reserved = ("v", "current_voltage", "is_spiking", "dt") if state == "t" else ("t", "dt")
    listed = _variables(sections.get("vars", []), reserved, bools)
    if state in bools:
        raise ModelError(f"'{state}' is a number")
    variables = [(n, d) for n, d in listed if n != state]
    if dict(listed).get(state, 0.0) != 0.0:
        raise ModelError(f"{what} {name}: '{state}' starts at 0 on the device (set the attribute after finalize)")
    if len(variables) > MAX_KINETICS_VARS:
        raise ModelError(f"more than {MAX_KINETICS_VARS} {what} variables")
    inputs = {"v", "is_spiking", "dt"} if state == "t" else {"t", "dt"}
    known = inputs | {state} | {n for n, _ in variables}
    all_bools = set(bools) | ({"is_spiking"} if state == "t" else set())

    def rename(n):
        n = "v" if (n == "current_voltage" and state == "t") else n
        if n not in known:
            raise ModelError(f"unknown variable {n!r}")
        return ("var", n)

    stmts = _convert_plain(_block(sections["on_iteration"])[0], rename, known - inputs, "on_iteration")
    _check_types(stmts, all_bools)
    return KineticsModel(name, state, variables, all_bools, stmts, dict(listed).get(state, 0.0))


def _parse_receptors(body):
    if not body or not body[0].startswith("type"):
        raise ModelError("[receptors]: 'type' comes first")
    m = re.match(r"^type\s*:\s*(.*)$", body[0])
    name = m.group(1).strip() if m else ""
    if not re.fullmatch(_NAME, name):
        raise ModelError(f"bad or missing type name {name!r}")
    groups, section = [{"name": None, "vars": [], "on_iteration": [], "states": None}], None
    kinetics = None
    for line in body[1:]:
        m = re.match(r"^(neurotransmitter|vars|on_iteration|receptors|kinetics)\s*:\s*(.*)$", line)
        if m and m.group(1) == "neurotransmitter":
            nt = m.group(2).strip()
            if not re.fullmatch(_NAME, nt):
                raise ModelError(f"bad neurotransmitter name {nt!r}")
            groups.append({"name": nt, "vars": [], "on_iteration": [], "states": None})
            section = None
        elif m and m.group(1) == "kinetics":              # single_kinetics_def, pest_ast/mod.rs:153
            if groups[-1]["name"] is not None or kinetics is not None:
                raise ModelError("[receptors]: one `kinetics: <type>` line, before the first neurotransmitter")
            kinetics = m.group(2).strip()
            if not re.fullmatch(_NAME, kinetics):
                raise ModelError(f"bad kinetics type {kinetics!r}")
            section = None
        elif m and m.group(1) == "receptors":             # receptor_vars_def, pest_ast/mod.rs:147: the type's receptor states
            if groups[-1]["name"] is None or groups[-1]["states"] is not None:
                raise ModelError("[receptors]: `receptors: <state>, ...` belongs to a neurotransmitter, once")
            states = [x.strip() for x in m.group(2).split(",") if x.strip()]
            if not states or any(not re.fullmatch(_NAME, x) for x in states) or len(set(states)) != len(states):
                raise ModelError(f"cannot read the receptor states {m.group(2)!r}")
            groups[-1]["states"] = states
            section = None
        elif m:
            section = m.group(1)
            if section == "on_iteration" and groups[-1]["name"] is None:
                raise ModelError("[receptors]: on_iteration belongs to a neurotransmitter")
            if m.group(2):
                groups[-1][section].append(m.group(2))
        elif section is None:
            raise ModelError(f"text outside a section: {line!r}")
        else:
            groups[-1][section].append(line)
    types = groups[1:]
    if not 1 <= len(types) <= 3:
        raise ModelError("[receptors] takes one to three neurotransmitter types (the exchange carries three)")
    if len({g["name"] for g in types}) != len(types):
        raise ModelError("[receptors]: a neurotransmitter is listed twice")
    model = ReceptorsModel(name, [], [], ())
    model.kinetics = kinetics
    # several states per type: every state (the lone `r` of a type without a `receptors:` line too) then lives in the
    # set's own variable table, see _finish_receptors
    model.multi = any(g["states"] is not None for g in types)
    model._groups = groups
    if not model.multi:
        _finish_receptors(model, None)
    return model


def _finish_receptors(model, receptor_kinetics):
    """Variable table and statements of a [receptors] block.  One state per type (the default): type k's `r` is the
    receptor state the handle keeps per neurotransmitter type (receptors$<Type>$r$kinetics$r) and the kinetics are the
    handle's.  Several states (`receptors: ampa_r, nmda_r`, lib.rs:7270-7316): each state is a value of the block's
    `kinetics:` type -- its `r` and the kinetics' own variables become the set's variables
    <Type>$<state>$kinetics$<var> -- and apply_r_change(t, dt) of a type runs the kinetics' on_iteration on each of its
    states in declaration order (model.kinetics_code[k])."""
    groups = model._groups
    types = groups[1:]
    bools, variables = set(), []
    reserved_top = ("v", "r", "dt", "t", "current")
    top = _variables(groups[0]["vars"], reserved_top, bools)
    variables += top
    top_names = {n for n, _ in top}
    if model.multi:
        if receptor_kinetics is None:
            kin_vars, kin_code, kin_bools = [], [("assign", "r", "=", ("var", "t"))], set()      # ApproximateReceptor: r = t
            if model.kinetics not in (None, "ApproximateReceptor"):
                raise ModelError(f"[receptors] {model.name}: kinetics type {model.kinetics!r} is not part of the description")
        else:
            if model.kinetics is not None and model.kinetics != receptor_kinetics.name:
                raise ModelError(f"[receptors] {model.name}: kinetics type {model.kinetics!r} is not part of the description")
            kin_vars, kin_code, kin_bools = receptor_kinetics.variables, receptor_kinetics.on_iteration, receptor_kinetics.bools
    out_types, kinetics_code, all_states = [], [], []
    for g in types:
        own_bools = set()
        states = (g["states"] or ["r"]) if model.multi else []
        own = _variables(g["vars"], ("v", "dt", "t") + (() if model.multi else ("r",)) + tuple(top_names), own_bools)
        if "current" in own_bools:
            raise ModelError("'current' is a number")
        own_names = {n for n, _ in own}
        if set(states) & (own_names | top_names):
            raise ModelError(f"[receptors] {g['name']}: a receptor state shares its name with a variable")
        variables += [(f"{g['name']}${n}", d) for n, d in own]
        bools |= {f"{g['name']}${n}" for n in own_bools}
        code = []
        for st in states:
            prefix = f"{g['name']}${st}$kinetics$"
            variables.append((prefix + "r", 0.0))
            variables += [(prefix + n, d) for n, d in kin_vars]
            bools |= {prefix + n for n in kin_bools if n != "r"}

            def kin_rename(n, prefix=prefix):
                return ("var", n) if n in ("t", "dt") else ("var", prefix + n)
            code += _rename_statements(kin_code, kin_rename)
        kinetics_code.append(code)
        all_states.append(states)
        if not g["on_iteration"]:
            raise ModelError(f"[receptors]: neurotransmitter {g['name']} has no on_iteration")

        def rename(n, g=g, own_names=own_names, states=states):
            if n in ("v", "current_voltage"):
                return ("var", "v")
            if n in states:
                return ("var", f"{g['name']}${n}$kinetics$r")
            if n == "r" and not model.multi:
                return ("var", "r")
            if n in own_names:
                return ("var", f"{g['name']}${n}")
            if n in top_names:
                return ("var", n)
            raise ModelError(f"[receptors] {g['name']}: unknown variable {n!r}")

        assignable = {f"{g['name']}${n}" for n in own_names} | top_names
        raw = _block(g["on_iteration"])[0]
        for st in _walk(raw):
            if st[0] == "diff":
                raise ModelError("[receptors]: on_iteration takes assignments, not differential equations")
        stmts = _convert_plain(raw, rename, assignable, "on_iteration")
        cur = f"{g['name']}$current"
        names = [n for n, _ in variables]
        out_types.append((g["name"], stmts, cur))
    if len(variables) > MAX_RECEPTOR_VARS:
        raise ModelError(f"more than {MAX_RECEPTOR_VARS} receptor variables")
    names = [n for n, _ in variables]
    model.types = [(nt, stmts, names.index(cur) if cur in names else None) for nt, stmts, cur in out_types]
    model.variables, model.bools = variables, set(bools)
    model.states, model.kinetics_code = all_states, kinetics_code
    for _, stmts, _ in model.types:
        _check_types(stmts, model.bools)
    for code in kinetics_code:
        _check_types(code, model.bools)
    return model


def _rename_statements(stmts, rename):
    """a copy of plain statements (assignments, [if]s, differential equations) with every variable renamed"""
    out = []
    for st in stmts:
        if st[0] == "if":
            out.append(("if", [(_map_expr(c, rename), _rename_statements(b, rename)) for c, b in st[1]],
                        None if st[2] is None else _rename_statements(st[2], rename)))
        else:
            out.append((st[0], rename(st[1])[1]) + tuple(st[2:-1]) + (_map_expr(st[-1], rename),))
    return out


def parse_description(text):
    """Every block of a description: [ion_channel]s, at most one [neuron], one [spike_train] and one
    [neural_refractoriness] (one generated library carries one of each)."""
    blocks = _split_blocks(text)
    channels = {}
    desc = Description()
    for kind, body in blocks:
        if kind == "ion_channel":
            ch = _parse_channel(body)
            if ch.name in channels:
                raise ModelError(f"ion channel {ch.name} is defined twice")
            channels[ch.name] = ch
        elif kind == "spike_train":
            if desc.spike_train is not None:
                raise ModelError("more than one [spike_train] block")
            desc.spike_train = _parse_spike_train(body)
        elif kind == "neural_refractoriness":
            if desc.refractoriness is not None:
                raise ModelError("more than one [neural_refractoriness] block")
            desc.refractoriness = _parse_refractoriness(body)
        elif kind == "neurotransmitter_kinetics":
            if desc.nt_kinetics is not None:
                raise ModelError("more than one [neurotransmitter_kinetics] block")
            desc.nt_kinetics = _parse_kinetics(body, "t")
        elif kind == "receptor_kinetics":
            if desc.receptor_kinetics is not None:
                raise ModelError("more than one [receptor_kinetics] block")
            desc.receptor_kinetics = _parse_kinetics(body, "r")
        elif kind == "receptors":
            if desc.receptors is not None:
                raise ModelError("more than one [receptors] block (one library carries one receptor set)")
            desc.receptors = _parse_receptors(body)
        elif kind != "neuron":
            raise ModelError(f"[{kind}] blocks are not supported")
    if desc.receptors is not None and desc.receptors.multi:
        _finish_receptors(desc.receptors, desc.receptor_kinetics)
    neurons = [body for kind, body in blocks if kind == "neuron"]
    if len(neurons) > 1:
        raise ModelError("expected exactly one [neuron] ... [end] block")
    if neurons:
        desc.neuron = _parse_neuron(neurons[0], channels)
        wanted = desc.neuron.receptors
        if wanted is not None and (desc.receptors is None or desc.receptors.name != wanted):
            raise ModelError(f"neuron {desc.neuron.name}: unknown receptors type {wanted!r}")
        if wanted is None and desc.receptors is not None:
            raise ModelError(f"[receptors] {desc.receptors.name} is not used: the neuron needs `receptors: "
                             f"{desc.receptors.name}`")
    elif desc.receptors is not None:
        raise ModelError("[receptors] blocks belong to a generated [neuron] (`receptors: <type>`); the built-in neurons "
                         "carry the ionotropic AMPA / NMDA / GABA set")
    elif channels:
        raise ModelError("[ion_channel] blocks without a [neuron] that uses them")
    if not desc.parts():
        raise ModelError("empty description")
    return desc


def parse(text):
    """Parse zero or more [ion_channel] blocks and ONE [neuron] block of the DSL subset in the module docstring."""
    desc = parse_description(text)
    if desc.neuron is None or [m for m in desc.parts() if m is not desc.receptors] != [desc.neuron]:
        raise ModelError("expected exactly one [neuron] ... [end] block (parse_description reads spike trains and "
                         "refractoriness)")
    return desc.neuron


def _parse_neuron(body, channels):
    sections, name = _sections(body, ("type", "vars", "on_spike", "spike_detection", "on_iteration",
                                      "on_electrochemical_iteration", "ion_channels", "receptors", "kinetics"))
    for need in ("on_iteration", "spike_detection"):
        if not sections.get(need):
            raise ModelError(f"section '{need}' is missing")
    model = NeuronModel(name, [], [], None, [])
    for var, value in _variables(sections.get("vars", []), ("v", "i", "is_spiking", "last_firing_time"), model.bools):
        if var in model.mandatory:
            if var in model.bools:
                raise ModelError(f"'{var}' is a number")
            model.mandatory[var] = value
        else:
            model.variables.append((var, value))
    instances = {}
    for item in ",".join(sections.get("ion_channels", [])).split(","):
        item = item.strip()
        if not item:
            continue
        m = re.fullmatch(rf"({_NAME})\s*=\s*({_NAME})", item)
        if not m:
            raise ModelError(f"cannot read ion channel {item!r}")
        inst, type_name = m.groups()
        if type_name not in channels:
            raise ModelError(f"unknown ion channel type {type_name!r}")
        if inst in instances or inst in dict(model.variables) or inst in model.mandatory or inst in ("v", "i"):
            raise ModelError(f"name {inst!r} is already taken")
        instances[inst] = channels[type_name]
        model.ion_channels.append((inst, type_name))
        model.variables += [(f"{inst}${f}", d) for f, d in channels[type_name].fields()]
        model.bools |= {f"{inst}${b}" for b in channels[type_name].bools}
    if len(model.variables) > MAX_VARS:
        raise ModelError(f"more than {MAX_VARS} variables")
    detect = " ".join(sections["spike_detection"]).strip()
    continuous = detect.replace(" ", "") == "continuous()"
    if continuous:
        # lib.rs:984-990 prints the peak detector of the built-in HodgkinHuxleyNeuron (hodgkin_huxley/mod.rs:207-220)
        # but forgets to define `last_voltage`; what is built here is that detector with last_voltage = the voltage at
        # the start of the iteration, exactly as the built-in neuron has it
        if "v_th" not in dict(model.variables):
            raise ModelError("continuous() compares with v_th: list it in vars")
        for hidden, is_bool in (("last_voltage", False), ("was_increasing", True)):
            if hidden in dict(model.variables):
                raise ModelError(f"continuous() keeps its own '{hidden}'")
            model.variables.append((hidden, 0.0))
            if is_bool:
                model.bools.add(hidden)
        detect = "v > v_th && was_increasing && !(last_voltage < v)"
    if len(model.variables) > MAX_VARS:
        raise ModelError(f"more than {MAX_VARS} variables")
    known = {"v", "i", "dt", "c_m", "gap_conductance"} | {n for n, _ in model.variables}

    def rename(n):                                  # c.current -> c$current
        key = n.replace(".", "$")
        if key not in known:
            raise ModelError(f"unknown variable {n!r}")
        return ("var", key)

    def convert(stmts, where):
        out = []
        for st in stmts:
            if st[0] == "if":
                out.append(("if", [(_map_expr(c, rename), convert(b, "an [if] branch")) for c, b in st[1]],
                            None if st[2] is None else convert(st[2], "an [if] branch")))
            elif st[0] == "struct_call" and where == "on_electrochemical_iteration" and \
                    st[1] in ("receptors", "synaptic_neurotransmitters"):
                # lib.rs:2275-2293: the calls the generated iterate_with_neurotransmitter_and_spike knows
                _, inst, method, args = st
                if (inst, method) == ("receptors", "update_receptor_kinetics"):
                    if args != [("var", "t"), ("var", "dt")]:
                        raise ModelError("receptors.update_receptor_kinetics takes (t, dt)")
                    out.append(("rc_update",))
                elif (inst, method) == ("receptors", "set_receptor_currents"):
                    if len(args) != 2 or args[1] != ("var", "dt"):
                        raise ModelError("receptors.set_receptor_currents takes (voltage, dt)")
                    out.append(("rc_set", _map_expr(args[0], rename)))
                elif (inst, method) == ("synaptic_neurotransmitters", "apply_t_changes"):
                    if args:
                        raise ModelError("synaptic_neurotransmitters.apply_t_changes takes no arguments")
                    out.append(("nt_apply",))
                else:
                    raise ModelError(f"cannot call {inst}.{method}()")
            elif st[0] == "struct_call":
                _, inst, method, args = st
                if inst not in instances or method != "update_current":
                    raise ModelError(f"cannot call {inst}.{method}(): only <ion channel>.update_current(...)")
                out.append(_inline_channel(inst, instances[inst], [_map_expr(a, rename) for a in args]))
            else:
                if st[0] == "diff" and where not in ("on_iteration", "on_electrochemical_iteration"):
                    raise ModelError(f"differential equations belong to the top level of on_iteration, not {where}")
                tgt = st[1].replace(".", "$")
                if tgt not in known - {"i"} or tgt in ("dt", "c_m", "gap_conductance"):
                    raise ModelError(f"cannot assign to {st[1]!r}")
                out.append((st[0], tgt) + tuple(st[2:-1]) + (_map_expr(st[-1], rename),))
        return out

    model.spike_detection = _map_expr(parse_expr(detect), rename)
    model.on_iteration = convert(_block(sections["on_iteration"])[0], "on_iteration")
    model.on_spike = convert(_block(sections.get("on_spike", []))[0], "on_spike")
    if continuous:
        remember = ("assign", "last_voltage", "=", ("var", "v"))
        model.on_iteration.insert(0, remember)
        model.after_detection = [("assign", "was_increasing", "=", ("bin", "<", ("var", "last_voltage"), ("var", "v")))]
    model.receptors = None
    if sections.get("receptors"):
        model.receptors = " ".join(sections["receptors"]).strip()
        if not re.fullmatch(_NAME, model.receptors):
            raise ModelError(f"bad receptors type {model.receptors!r}")
    model.on_electrochemical_iteration = None
    if sections.get("on_electrochemical_iteration"):
        model.on_electrochemical_iteration = convert(_block(sections["on_electrochemical_iteration"])[0],
                                                     "on_electrochemical_iteration")
        if continuous:
            model.on_electrochemical_iteration.insert(0, remember)
    for where, stmts in (("on_iteration", model.on_iteration), ("on_spike", model.on_spike)):
        for st in _walk(stmts):
            exprs = [c for c, _ in st[1]] if st[0] == "if" else [st[-1]] if st[0] in ("diff", "assign") else []
            if any(_has_rc_get(e) for e in exprs):
                raise ModelError(f"receptors.get_receptor_currents belongs to on_electrochemical_iteration, not {where}")
    if _has_rc_get(model.spike_detection):
        raise ModelError("receptors.get_receptor_currents belongs to on_electrochemical_iteration, not spike_detection")
    _check_types(model.on_iteration + model.on_spike + (model.on_electrochemical_iteration or []) + model.after_detection,
                 model.bools, condition=model.spike_detection)
    return model


def _has_rc_get(e):
    if e[0] == "rc_get":
        return True
    return any(_has_rc_get(sub) for sub in e[1:] if isinstance(sub, tuple)) or \
        any(_has_rc_get(x) for sub in e[1:] if isinstance(sub, list) for x in sub)


def _check_types(stmts, bools, condition=None, number=None):
    """The generated Rust is typed (f32 / bool fields): a description that rustc would refuse is refused here."""
    def kind(e):
        k = e[0]
        if k == "num":
            return "number"
        if k == "bool":
            return "bool"
        if k == "var":
            return "bool" if e[1] in bools else "number"
        if k == "neg" or k == "powi":
            need(e[1], "number", "arithmetic")
            return "number"
        if k == "rc_get":
            need(e[1], "number", "get_receptor_currents")
            need(e[2], "number", "get_receptor_currents")
            return "number"
        if k == "not":
            need(e[1], "bool", "'!'")
            return "bool"
        if k == "call":
            for a in e[2]:
                need(a, "number", f"{e[1]}()")
            return "bool" if e[1] == "isnan" else "number"
        _, op, lhs, rhs = e
        if op in ("&&", "||"):
            need(lhs, "bool", f"'{op}'")
            need(rhs, "bool", f"'{op}'")
            return "bool"
        if op in ("==", "!="):
            if kind(lhs) != kind(rhs):
                raise ModelError(f"'{op}' compares a number with a bool")
            return "bool"
        need(lhs, "number", f"'{op}'")
        need(rhs, "number", f"'{op}'")
        return "bool" if op in (">=", "<=", ">", "<") else "number"

    def need(e, want, where):
        got = kind(e)
        if got != want:
            raise ModelError(f"{where} needs a {want}, not a {got}")

    if condition is not None:
        need(condition, "bool", "spike_detection")
    if number is not None:
        need(number, "number", "effect")
    for st in _walk(stmts):
        if st[0] == "if":
            for cond, _ in st[1]:
                need(cond, "bool", "[if]")
        elif st[0] == "diff":
            if st[1] in bools:
                raise ModelError(f"d{st[1]}/dt: {st[1]} is a bool")
            need(st[2], "number", "a differential equation")
        elif st[0] == "rc_set":
            need(st[1], "number", "set_receptor_currents")
        elif st[0] == "assign":
            want = "bool" if st[1] in bools else "number"
            if want == "bool" and st[2] != "=":
                raise ModelError(f"'{st[2]}' on the bool variable {st[1]}")
            need(st[3], want, f"assignment to {st[1].replace('$', '.')}")


# ---- HIP -------------------------------------------------------------------------------------------------
def _f32_literal(x):
    y = struct.unpack("f", struct.pack("f", x))[0]
    return repr(y) + ("f" if ("." in repr(y) or "e" in repr(y) or "inf" in repr(y)) else ".0f")


def _hip_expr(e, index):
    kind = e[0]
    if kind == "num":
        return _f32_literal(e[1])
    if kind == "bool":
        return "true" if e[1] else "false"
    if kind == "var":
        if e[1] in index["$base"]:
            return index["$base"][e[1]]
        if e[1] in index.get("$bools", ()):
            return f"(x[{index[e[1]]}] != 0.0f)"
        return f"x[{index[e[1]]}]"
    if kind == "neg":
        return f"(-{_hip_expr(e[1], index)})"
    if kind == "not":
        return f"(!{_hip_expr(e[1], index)})"
    if kind == "call":
        if e[1] == "isnan":
            arg = _hip_expr(e[2][0], index)
            return f"({arg} != {arg})"
        fn = {"exp": "expf_glibc", "tanh": "tanhf_portable", "sinh": "sinhf_portable", "cosh": "coshf_portable",
              "sin": "sinf_portable", "cos": "cosf_portable", "tan": "tanf_portable",
              "heaviside": "heaviside_rs", "min": "min_rs", "max": "max_rs", "powf": "powf_glibc",
              "rpow": "rpowf_glibc"}[e[1]]
        return f"{fn}({', '.join(_hip_expr(a, index) for a in e[2])})"
    if kind == "powi":
        return f"powif_glibc({_hip_expr(e[1], index)}, {e[2]})"
    if kind == "rc_get":
        return f"chem.get_receptor_currents({_hip_expr(e[1], index)}, {_hip_expr(e[2], index)})"
    _, op, lhs, rhs = e
    return f"({_hip_expr(lhs, index)} {op} {_hip_expr(rhs, index)})"


def _hip_statements(stmts, index, with_diffs, indent="    "):
    lines, diffs = [], []
    for s in stmts:
        if s[0] == "if":
            for k, (cond, body) in enumerate(s[1]):
                lines.append(f"{indent}{'if' if k == 0 else '} else if'} ({_hip_expr(cond, index)}) {{")
                lines.append(_hip_statements(body, index, False, indent + "    "))
            if s[2] is not None:
                lines.append(f"{indent}}} else {{")
                lines.append(_hip_statements(s[2], index, False, indent + "    "))
            lines.append(f"{indent}}}")
            continue
        if s[0] == "scope":                 # an inlined ion channel: its `x += dx` at the end of its own body
            lines.append(f"{indent}{{")
            lines.append(_hip_statements(s[1], index, True, indent + "    "))
            lines.append(f"{indent}}}")
            continue
        if s[0] == "rc_update":
            lines.append(f"{indent}chem.update_receptor_kinetics();")
            continue
        if s[0] == "rc_set":
            lines.append(f"{indent}chem.set_receptor_currents({_hip_expr(s[1], index)});")
            continue
        if s[0] == "nt_apply":
            lines.append(f"{indent}chem.apply_t_changes(v);")
            continue
        target = index["$base"][s[1]] if s[1] in index["$base"] else f"x[{index[s[1]]}]"
        if s[0] == "diff":
            d = f"d_{s[1]}" if s[1] in index["$base"] else f"d_x{index[s[1]]}"
            lines.append(f"{indent}const float {d} = ({_hip_expr(s[2], index)}) * dt;")
            diffs.append(f"{indent}{target} += {d};")
        elif s[1] in index.get("$bools", ()) and s[1] not in index["$base"]:
            lines.append(f"{indent}{target} = {_hip_expr(s[3], index)} ? 1.0f : 0.0f;")
        else:
            lines.append(f"{indent}{target} {s[2]} {_hip_expr(s[3], index)};")
    return "\n".join(l for l in lines + (diffs if with_diffs else []) if l)


_NEURON_BASE = {"v": "v", "i": "i_in", "dt": "dt", "c_m": "c_m", "gap_conductance": "g_gap"}
_ST_BASE = {"v": "v", "is_spiking": "is_spiking", "dt": "dt", "v_resting": "v_resting", "v_th": "v_th"}
_REFR_BASE = {"time_difference": "time_difference", "v_th": "v_th", "v_resting": "v_resting", "dt": "dt", "decay": "decay"}


def _table(variables):
    names = ", ".join(f'"{n}"' for n, _ in variables) or '""'
    defaults = ", ".join(_f32_literal(d) for _, d in variables) or "0.0f"
    return f"""constexpr int NVARS = {len(variables)};
constexpr int NSTORE = {max(1, len(variables))};
static const char *const NAMES[NSTORE] = {{{names}}};
static const float DEFAULTS[NSTORE] = {{{defaults}}};"""


def _neuron_source(model):
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    index["$bools"] = model.bools
    index["$base"] = _NEURON_BASE
    m = model.mandatory
    return f"""namespace custom {{
static const char *const TYPE_NAME = "{model.name}";
{_table(model.variables)}
constexpr float DEFAULT_VOLTAGE = {_f32_literal(m['current_voltage'])}, DEFAULT_DT = {_f32_literal(m['dt'])},
                DEFAULT_C_M = {_f32_literal(m['c_m'])}, DEFAULT_GAP = {_f32_literal(m['gap_conductance'])};

__device__ __forceinline__ void on_iteration(float &v, float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
{_hip_statements(model.on_iteration, index, True)}
}}
__device__ __forceinline__ bool spike_detection(float v, float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
    const bool spiking = {_hip_expr(model.spike_detection, index)};
{_hip_statements(model.after_detection, index, False)}
    return spiking;
}}
__device__ __forceinline__ void on_spike(float &v, float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
{_hip_statements(model.on_spike, index, False)}
}}
// on_electrochemical_iteration (nb_macro lib.rs:2280-2316): `chem` = the receptors / transmitters of this neuron
constexpr bool HAS_ELECTROCHEMICAL = {'true' if model.on_electrochemical_iteration is not None else 'false'};
template <class Chem>
__device__ __forceinline__ void on_electrochemical_iteration(float &v, float (&x)[NSTORE], float i_in, float dt, float c_m,
                                                             float g_gap, Chem &chem)
{{
{_hip_statements(model.on_electrochemical_iteration or [], index, True)}
}}
}} // namespace custom
"""


def _spike_train_source(model):
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    index["$bools"] = model.bools
    index["$base"] = _ST_BASE
    m = model.mandatory
    return f"""#define SNN_HAVE_CUSTOM_SPIKE_TRAIN 1
namespace custom_st {{
static const char *const TYPE_NAME = "{model.name}";
{_table(model.variables)}
constexpr float DEFAULT_VOLTAGE = {_f32_literal(m['current_voltage'])}, DEFAULT_DT = {_f32_literal(m['dt'])},
                DEFAULT_V_RESTING = {_f32_literal(m['v_resting'])}, DEFAULT_V_TH = {_f32_literal(m['v_th'])};

// iterate() of the generated spike train (nb_macro lib.rs:4884-4891) up to the neurotransmitter update
__device__ __forceinline__ void on_iteration(float &v, bool &is_spiking, float (&x)[NSTORE], float dt, float v_resting,
                                             float v_th)
{{
{_hip_statements(model.on_iteration, index, True)}
}}
}} // namespace custom_st
"""


def _refractoriness_source(model):
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    index["$bools"] = set()
    index["$base"] = _REFR_BASE
    return f"""#define SNN_HAVE_CUSTOM_REFRACTORINESS 1
namespace custom_refr {{
static const char *const TYPE_NAME = "{model.name}";
{_table(model.variables)}
constexpr float DEFAULT_DECAY = {_f32_literal(model.decay)};

// get_effect of the generated NeuralRefractoriness (nb_macro lib.rs:5736-5750)
__device__ __forceinline__ float effect(float time_difference, float v_th, float v_resting, float dt, float decay,
                                        const float (&x)[NSTORE])
{{
    return {_hip_expr(model.effect, index)};
}}
}} // namespace custom_refr
"""


def _kinetics_source(model):
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    index["$bools"] = model.bools
    if model.state == "t":
        index["$base"] = {"t": "t", "v": "v", "is_spiking": "is_spiking", "dt": "dt"}
        ns, macro, args = "custom_nt", "SNN_HAVE_CUSTOM_NT", "float &t, float (&x)[NSTORE], float v, bool is_spiking, float dt"
        cite = "apply_t_change of the generated NeurotransmitterKinetics (nb_macro lib.rs:6489-6498)"
    else:
        index["$base"] = {"r": "r", "t": "t", "dt": "dt"}
        ns, macro, args = "custom_rc", "SNN_HAVE_CUSTOM_RC", "float &r, float (&x)[NSTORE], float t, float dt"
        cite = "apply_r_change of the generated ReceptorKinetics (nb_macro lib.rs:6778-6786)"
    return f"""#define {macro} 1
namespace {ns} {{
static const char *const TYPE_NAME = "{model.name}";
{_table(model.variables)}
constexpr float DEFAULT_STATE = {_f32_literal(model.state_default)};

// {cite}
__device__ __forceinline__ void apply({args})
{{
{_hip_statements(model.on_iteration, index, True)}
}}
}} // namespace {ns}
"""


def _receptors_source(model):
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    index["$bools"] = model.bools
    index["$base"] = {"v": "v", "r": "r"}
    bodies = []
    for k, (nt, stmts, _) in enumerate(model.types):
        bodies.append(f"    {'if' if k == 0 else '} else if'} (k == {k}) {{        // {nt}\n"
                      + _hip_statements(stmts, index, False, "        "))
    kin = []
    kin_index = dict(index)
    kin_index["$base"] = {"t": "t", "dt": "dt"}
    for k, code in enumerate(model.kinetics_code if model.multi else []):
        kin.append(f"    {'if' if k == 0 else '} else if'} (k == {k}) {{        // {model.types[k][0]}: {', '.join(model.states[k])}\n"
                   + _hip_statements(code, kin_index, True, "        "))
    kin_body = (chr(10).join(kin) + "\n    }") if kin else "    (void)k; (void)t; (void)dt; (void)x;"
    nt_names = ", ".join(f'"{t[0]}"' for t in model.types) + ', ""' * (3 - len(model.types))
    cur = ", ".join(str(-1 if t[2] is None else t[2]) for t in model.types) + ", -1" * (3 - len(model.types))
    return f"""#define SNN_HAVE_CUSTOM_RECEPTORS 1
namespace custom_receptors {{
static const char *const TYPE_NAME = "{model.name}";
constexpr int NTYPES = {len(model.types)};
static const char *const NT_NAMES[3] = {{{nt_names}}};
{_table(model.variables)}
constexpr int CURRENT_INDEX[3] = {{{cur}}};          // the type's `current` in x, -1: the type carries no current

// several receptor states per type (`receptors: a, b`): every state's r and kinetics variables are among x, and
// <Type>Receptor::apply_r_change (lib.rs:7306-7316) -- the kinetics' on_iteration on each state of type k -- is here
constexpr bool MULTI_STATE = {'true' if model.multi else 'false'};
__device__ __forceinline__ void update_kinetics(int k, float t, float dt, float (&x)[NSTORE])
{{
{kin_body}
}}

// <Type>Receptor::iterate of the generated receptor set (nb_macro lib.rs:7296-7340): type k's on_iteration
__device__ __forceinline__ void iterate(int k, float v, float r, float (&x)[NSTORE])
{{
{chr(10).join(bodies)}
    }}
}}
}} // namespace custom_receptors
"""


def hip_source(model):
    """The generated header for a NeuronModel or a Description: per block a variable table and its code as device
    functions (namespaces custom / custom_st / custom_refr)."""
    desc = model if isinstance(model, Description) else Description(neuron=model)
    parts = []
    if desc.neuron is not None:
        parts.append("#define SNN_HAVE_CUSTOM_NEURON 1\n" + _neuron_source(desc.neuron))
    if desc.spike_train is not None:
        parts.append(_spike_train_source(desc.spike_train))
    if desc.refractoriness is not None:
        parts.append(_refractoriness_source(desc.refractoriness))
    if desc.nt_kinetics is not None:
        parts.append(_kinetics_source(desc.nt_kinetics))
    if desc.receptor_kinetics is not None:
        parts.append(_kinetics_source(desc.receptor_kinetics))
    if desc.receptors is not None:
        parts.append(_receptors_source(desc.receptors))
    return f"""// GENERATED by spiking-neural-networks_amd/modelgen.py from the description of {desc.name}
// (nb_macro semantics, see modelgen.py).  Included through csrc/snn_custom_model.hpp.
#pragma once
namespace snn {{
{chr(10).join(parts)}}} // namespace snn
"""
