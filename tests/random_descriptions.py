"""Random neuron descriptions for property tests of the description generator (test infrastructure): `count` variables
f0.. each assigned a random expression over the input current, three heterogeneous parameters and the earlier
results -- every operator, function and an if / else per statement."""
import numpy as np

UNARY = ("exp", "tanh", "sinh", "cosh", "sin", "cos", "tan", "heaviside")
BINARY_FUNCTIONS = ("min", "max")
COMPARISONS = ("<", "<=", ">", ">=", "==", "!=")


def expression(rng, names, depth):
    roll = rng.random()
    if depth == 0 or roll < 0.18:
        if rng.random() < 0.35:
            return repr(round(float(rng.uniform(-3, 3)), 3)).replace("-", "-")
        return str(rng.choice(names))
    if roll < 0.62:
        op = rng.choice(["+", "-", "*", "/"])
        return f"({expression(rng, names, depth - 1)} {op} {expression(rng, names, depth - 1)})"
    if roll < 0.70:
        return f"(-{expression(rng, names, depth - 1)})"
    if roll < 0.84:
        fn = str(rng.choice(UNARY))
        inner = expression(rng, names, depth - 1)
        if fn in ("exp", "sinh", "cosh"):
            inner = f"min({inner}, 6)"               # keep the magnitudes finite often enough to be interesting
        return f"{fn}({inner})"
    if roll < 0.92:
        fn = str(rng.choice(BINARY_FUNCTIONS))
        return f"{fn}({expression(rng, names, depth - 1)}, {expression(rng, names, depth - 1)})"
    return f"({expression(rng, names, depth - 1)}) ^ {int(rng.choice([2, 3, -1, -2, 4]))}"


def condition(rng, names):
    c = f"{expression(rng, names, 2)} {rng.choice(COMPARISONS)} {expression(rng, names, 2)}"
    roll = rng.random()
    if roll < 0.25:
        c = f"{c} && {expression(rng, names, 1)} {rng.choice(COMPARISONS)} {expression(rng, names, 1)}"
    elif roll < 0.5:
        c = f"{c} || !({expression(rng, names, 1)} {rng.choice(COMPARISONS)} {expression(rng, names, 1)})"
    elif roll < 0.6:
        c = f"isnan({expression(rng, names, 2)}) || {c}"
    return c


def description(seed, count=24, name="RandomExpressions"):
    rng = np.random.default_rng(seed)
    names = ["i", "a", "b", "c"]
    lines = []
    for k in range(count):
        target = f"f{k}"
        if rng.random() < 0.3:
            lines += [f"        [if] {condition(rng, names)} [then]",
                      f"            {target} = {expression(rng, names, 3)}",
                      "        [else]",
                      f"            {target} = {expression(rng, names, 3)}",
                      "        [end]"]
        else:
            lines.append(f"        {target} = {expression(rng, names, 4)}")
        names.append(target)
    variables = ", ".join(["v_th = 50000000", "a = 0.7", "b = -1.3", "c = 2.1"] + [f"f{k} = 0" for k in range(count)])
    return "\n".join(["[neuron]", f"    type: {name}", f"    vars: {variables}", "    spike_detection: v >= v_th",
                      "    on_iteration:"] + lines + ["[end]"])
