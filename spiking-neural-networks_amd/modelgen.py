"""Neuron-model description -> HIP: a first slice of the reference's `neuron_builder!` DSL
(/root/reference/build_test/nb_macro/src/lib.rs; grammar build_test/nb_macro/src/ast.pest) for integrate-and-fire
models -- one `[neuron]` block with `type`, `vars`, `on_iteration`, `spike_detection` and `on_spike`:

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
  * electrical step: on_iteration; neurotransmitter release; is_spiking = spike_detection; on_spike if spiking;
    with neurotransmission: receptor kinetics and currents at the old voltage first, and
    `v -= receptor currents * (dt / c_m)` right after on_iteration (the Ionotropic AMPA/NMDA/GABA receptors of the
    hot path stand where the generated Rust uses its receptor type);
  * every binary operation is one float32 operation, evaluated left to right as written (no contraction).

  * `[if] cond [then] ... [elseif] cond [then] ... [else] ... [end]` (nestable, lib.rs:405-470) in on_iteration and
    on_spike; differential equations stay at the top level of on_iteration (a branch-local `dx` would be out of
    scope where the generated Rust applies it).

`hip_source(model)` emits the header that csrc/snn_custom_model.hpp includes when the library is compiled with
-DSNN_CUSTOM_MODEL_HEADER; `_lib.build_custom(model)` compiles such a library.  Not supported (rejected with a
message): ion channels, receptors / kinetics blocks, bool variables, `^`, functions other than exp,
`continuous()` spike detection, on_electrochemical_iteration.
"""
import re
import struct

MANDATORY = {"current_voltage": 0.0, "dt": 0.1, "c_m": 1.0, "gap_conductance": 10.0}
MAX_VARS = 16


class ModelError(ValueError):
    pass


# ---- tokens -----------------------------------------------------------------------------------------
_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?)|([A-Za-z_][A-Za-z_0-9]*)|"
                    r"(\|\||&&|==|!=|>=|<=|[-+*/()<>!,^]))")


def _tokens(text):
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ModelError(f"cannot read expression at: {text[pos:]!r}")
        num, name, op = m.groups()
        out.append(("num", num) if num else ("name", name) if name else ("op", op))
        pos = m.end()
    return out


# ---- expressions: ("num", f) | ("var", name) | ("neg", e) | ("not", e) | ("bin", op, l, r) | ("call", name, [args])
_LEVELS = [("||",), ("&&",), ("==", "!=", ">=", "<=", ">", "<"), ("+", "-"), ("*", "/")]


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
            lhs = ("bin", op, lhs, self.expr(level + 1))
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
                if val != "exp" or len(args) != 1:
                    raise ModelError(f"function {val}() is not supported (only exp(x))")
                return ("call", val, args)
            return ("var", val)
        if (kind, val) == ("op", "("):
            e = self.expr()
            if self.take() != ("op", ")"):
                raise ModelError("missing ')'")
            return e
        if (kind, val) == ("op", "^"):
            raise ModelError("'^' is not supported")
        raise ModelError(f"unexpected token {val!r}")


def parse_expr(text):
    p = _Parser(_tokens(text))
    e = p.expr()
    if p.peek() != (None, None):
        if p.peek() == ("op", "^"):
            raise ModelError("'^' is not supported")
        raise ModelError(f"trailing input in expression {text!r}")
    return e


# ---- statements: ("diff", name, expr) | ("assign", name, op, expr) -------------------------------------
_DIFF = re.compile(r"^d([A-Za-z_][A-Za-z_0-9]*)\s*/\s*dt\s*=\s*(.+)$")
_ASSIGN = re.compile(r"^([A-Za-z_][A-Za-z_0-9]*)\s*(=|\+=|-=|\*=|/=)\s*(.+)$")


def _statement(line):
    m = _DIFF.match(line)
    if m:
        return ("diff", m.group(1), parse_expr(m.group(2)))
    m = _ASSIGN.match(line)
    if m:
        return ("assign", m.group(1), m.group(2), parse_expr(m.group(3)))
    raise ModelError(f"cannot read statement {line!r} (struct calls are not supported)")


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
        self.variables = variables          # [(name, default)] in declaration order, without the mandatory ones
        self.mandatory = dict(MANDATORY)    # defaults of current_voltage / dt / c_m / gap_conductance
        self.on_iteration, self.spike_detection, self.on_spike = on_iteration, spike_detection, on_spike


def parse(text):
    """Parse ONE [neuron] block of the DSL subset described in the module docstring."""
    lines_all = [l.strip() for l in text.splitlines() if l.strip()]
    heads = [l for l in lines_all if re.fullmatch(r"\[(neuron|ion_channel|receptors|neurotransmitter_kinetics|"
                                                  r"receptor_kinetics|spike_train|neural_refractoriness)\]", l)]
    if heads != ["[neuron]"] or lines_all[0] != "[neuron]":
        raise ModelError("expected exactly one [neuron] ... [end] block (ion channels, receptors, kinetics and spike "
                         "trains are not supported)")
    depth, body = 0, None
    for k, l in enumerate(lines_all[1:], start=1):
        if _IF.match(l) and l.startswith("[if]"):
            depth += 1
        elif l == "[end]":
            if depth == 0:
                body = lines_all[1:k]
                if k != len(lines_all) - 1:
                    raise ModelError("text after the [neuron] block")
                break
            depth -= 1
    if body is None:
        raise ModelError("[neuron] without [end]")
    sections, current = {}, None
    for line in body:
        m = re.match(r"^(type|vars|on_spike|spike_detection|on_iteration|on_electrochemical_iteration|ion_channels|"
                     r"kinetics|receptors)\s*:\s*(.*)$", line)
        if m:
            current = m.group(1)
            sections.setdefault(current, [])
            if m.group(2):
                sections[current].append(m.group(2))
        elif current is None:
            raise ModelError(f"text outside a section: {line!r}")
        else:
            sections[current].append(line)
    for bad in ("on_electrochemical_iteration", "ion_channels", "kinetics", "receptors"):
        if bad in sections:
            raise ModelError(f"section '{bad}' is not supported")
    for need in ("type", "on_iteration", "spike_detection"):
        if not sections.get(need):
            raise ModelError(f"section '{need}' is missing")
    name = sections["type"][0].strip()
    if not re.fullmatch(r"[A-Za-z_][A-Za-z_0-9]*", name):
        raise ModelError(f"bad type name {name!r}")
    model = NeuronModel(name, [], [], None, [])
    for item in ",".join(sections.get("vars", [])).split(","):
        item = item.strip()
        if not item:
            continue
        m = re.fullmatch(r"([A-Za-z_][A-Za-z_0-9]*)\s*=\s*(-?\s*[0-9.eE+-]+|true|false)", item)
        if not m:
            raise ModelError(f"cannot read variable {item!r}")
        if m.group(2) in ("true", "false"):
            raise ModelError("bool variables are not supported")
        var, value = m.group(1), float(m.group(2).replace(" ", ""))
        if var in ("v", "i", "is_spiking", "last_firing_time"):
            raise ModelError(f"'{var}' is reserved")
        if var in model.mandatory:
            model.mandatory[var] = value
        elif var in dict(model.variables):
            raise ModelError(f"variable {var} is defined twice")
        else:
            model.variables.append((var, value))
    if len(model.variables) > MAX_VARS:
        raise ModelError(f"more than {MAX_VARS} variables")
    detect = " ".join(sections["spike_detection"]).strip()
    if detect.replace(" ", "") == "continuous()":
        raise ModelError("continuous() spike detection is not supported")
    model.spike_detection = parse_expr(detect)
    model.on_iteration = _block(sections["on_iteration"])[0]
    model.on_spike = _block(sections.get("on_spike", []))[0]

    def no_diffs(stmts, where):
        for s in stmts:
            if s[0] == "diff":
                raise ModelError(f"differential equations belong to the top level of on_iteration, not {where}")
            if s[0] == "if":
                for _, body_ in s[1]:
                    no_diffs(body_, "an [if] branch")
                if s[2] is not None:
                    no_diffs(s[2], "an [if] branch")
    no_diffs(model.on_spike, "on_spike")
    for s in model.on_iteration:
        if s[0] == "if":
            no_diffs([s], "an [if] branch")
    known = {"v", "i", "dt", "c_m", "gap_conductance"} | {n for n, _ in model.variables}

    def check(e):
        if e[0] == "var" and e[1] not in known:
            raise ModelError(f"unknown variable {e[1]!r}")
        for sub in e[1:]:
            if isinstance(sub, tuple):
                check(sub)
            elif isinstance(sub, list):
                for x in sub:
                    check(x)
    def check_block(stmts):
        for s in stmts:
            if s[0] == "if":
                for cond, body_ in s[1]:
                    check(cond)
                    check_block(body_)
                if s[2] is not None:
                    check_block(s[2])
                continue
            if s[1] not in known - {"i"} or s[1] in ("dt", "c_m", "gap_conductance"):
                raise ModelError(f"cannot assign to {s[1]!r}")
            check(s[-1])
    check_block(model.on_iteration + model.on_spike)
    check(model.spike_detection)
    return model


# ---- HIP -------------------------------------------------------------------------------------------------
def _f32_literal(x):
    y = struct.unpack("f", struct.pack("f", x))[0]
    return repr(y) + ("f" if ("." in repr(y) or "e" in repr(y) or "inf" in repr(y)) else ".0f")


def _hip_expr(e, index):
    kind = e[0]
    if kind == "num":
        return _f32_literal(e[1])
    if kind == "var":
        if e[1] in ("v", "i", "dt", "c_m", "gap_conductance"):
            return {"v": "v", "i": "i_in", "dt": "dt", "c_m": "c_m", "gap_conductance": "g_gap"}[e[1]]
        return f"x[{index[e[1]]}]"
    if kind == "neg":
        return f"(-{_hip_expr(e[1], index)})"
    if kind == "not":
        return f"(!{_hip_expr(e[1], index)})"
    if kind == "call":
        return f"expf_portable({_hip_expr(e[2][0], index)})"
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
        target = "v" if s[1] == "v" else f"x[{index[s[1]]}]"
        if s[0] == "diff":
            lines.append(f"{indent}const float d_{s[1]} = ({_hip_expr(s[2], index)}) * dt;")
            diffs.append(f"{indent}{target} += d_{s[1]};")
        else:
            lines.append(f"{indent}{target} {s[2]} {_hip_expr(s[3], index)};")
    return "\n".join(l for l in lines + (diffs if with_diffs else []) if l)


def hip_source(model):
    """The generated header: variable table + on_iteration / spike_detection / on_spike as device functions."""
    index = {n: k for k, (n, _) in enumerate(model.variables)}
    nv = max(1, len(model.variables))
    names = ", ".join(f'"{n}"' for n, _ in model.variables) or '""'
    defaults = ", ".join(_f32_literal(d) for _, d in model.variables) or "0.0f"
    m = model.mandatory
    return f"""// GENERATED by spiking-neural-networks_amd/modelgen.py from the neuron description of type {model.name}
// (nb_macro semantics, see modelgen.py).  Included through csrc/snn_custom_model.hpp.
#pragma once
namespace snn {{
namespace custom {{
constexpr int NVARS = {len(model.variables)};
constexpr int NSTORE = {nv};
static const char *const TYPE_NAME = "{model.name}";
static const char *const NAMES[NSTORE] = {{{names}}};
static const float DEFAULTS[NSTORE] = {{{defaults}}};
constexpr float DEFAULT_VOLTAGE = {_f32_literal(m['current_voltage'])}, DEFAULT_DT = {_f32_literal(m['dt'])},
                DEFAULT_C_M = {_f32_literal(m['c_m'])}, DEFAULT_GAP = {_f32_literal(m['gap_conductance'])};

__device__ __forceinline__ void on_iteration(float &v, float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
{_hip_statements(model.on_iteration, index, True)}
}}
__device__ __forceinline__ bool spike_detection(float v, const float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
    return {_hip_expr(model.spike_detection, index)};
}}
__device__ __forceinline__ void on_spike(float &v, float (&x)[NSTORE], float i_in, float dt, float c_m, float g_gap)
{{
{_hip_statements(model.on_spike, index, False)}
}}
}} // namespace custom
}} // namespace snn
"""
