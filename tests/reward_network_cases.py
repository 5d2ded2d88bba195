"""Random RewardModulatedLatticeNetwork inputs (neuron/mod.rs:3419-3453) for the connection-kind tests: lattices that are
reward-modulated, plain with STDP ("plastic") or plain without ("fixed"), spike-train cells, and a RewardModulatedConnection kind
per pair of lattices drawn from what the reference defines for that pair (snn_o_reward_cross_check) -- plus, on request, one
deliberate violation of each refusal class.  Inputs only: nothing is stepped here."""
import numpy as np

import oracle_binding as ob
import parity

MOD, PLASTIC, FIXED = 0, 1, 2
# kinds the reference defines between two lattice roles (0 = left to the plain network's rule)
ALLOWED = {(MOD, MOD): (0, 1, 1, 2), (MOD, PLASTIC): (0, 1, 1), (MOD, FIXED): (0, 1, 2), (PLASTIC, PLASTIC): (0, 2, 2),
           (PLASTIC, FIXED): (0, 2, 2), (FIXED, FIXED): (0, 1, 2)}
FROM_CELLS = {MOD: (0, 1, 1, 2), PLASTIC: (0, 2), FIXED: (0, 1, 2)}


def draw(seed, violation=0, paused=False):
    """paused: one of the reward-modulated lattices has do_modulation switched off -- it stays a reward-modulated lattice for its
    partners' visits (which map of the reference's network holds it), takes every reward, but is never visited itself"""
    rng = np.random.default_rng(4000 + seed)
    n_lat = int(rng.integers(2, 5))
    lattices = [(2 * i + int(rng.integers(0, 2)), int(rng.integers(1, 6)), int(rng.integers(1, 7))) for i in range(n_lat)]
    st = [(100 + i, int(rng.integers(2, 4)), int(rng.integers(2, 5))) for i in range(int(rng.integers(0, 3)) if seed % 4 == 3 else int(rng.integers(1, 3)))]
    st_kind = ob.ST_POISSON if st else ob.ST_NONE
    net = parity.make_oracle(parity.Layout(lattices, st), st_kind=st_kind, model=[ob.IZHIKEVICH, ob.LIF][seed % 2])
    nn, nc = net.n_neurons, net.n_cells
    lo, hi = ((-65, 30) if seed % 2 == 0 else (-80, -50))
    net["current_voltage"] = ob.uniform_array(seed, nn, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, nn, 8.0, 14.0)
    if seed % 2:
        net["tref"] = ob.uniform_array(seed + 2, nn, 0.2, 2.0)
        net["tau_m"] = 10.0
    if nc:
        net["st_chance_of_firing"] = ob.uniform_array(seed + 4, nc, 0.1, 0.5)
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
    role = rng.integers(0, 3, n_lat)
    role[int(rng.integers(0, n_lat))] = MOD                             # at least one modulated lattice
    if violation == 2:
        role[:2] = (PLASTIC, FIXED)
    if violation == 3:
        role[:2] = (PLASTIC, MOD)
    if violation in (1, 4):
        role[:2] = (PLASTIC, MOD) if violation == 1 else (PLASTIC, FIXED)
    for slot in range(n_lat):
        if role[slot] == MOD:
            net["rm_do_modulation"][slot] = 1
            net["rm_dopamine"][slot] = float(rng.uniform(-0.01, 0.02))
            net["rm_tau_d"][slot] = float(rng.uniform(2.0, 10.0))
            net["rm_tau_c"][slot] = float(rng.uniform(0.02, 0.2))
            net["rm_a_plus"][slot] = float(rng.uniform(0.001, 0.01))
            net["rm_a_minus"][slot] = float(rng.uniform(0.001, 0.01))
        else:
            net["do_plasticity"][slot] = int(role[slot] == PLASTIC)
            net["stdp_a_plus"][slot] = float(rng.uniform(0.5, 2.5))
            net["stdp_tau_minus"][slot] = float(rng.uniform(2.0, 6.0))
    # graph: random, then every connection between two lattices gets its reverse (the outgoing halves look it up) -- except some
    # connections OUT of fixed lattices, whose neurons are never visited
    net.fill_graph(seed + 6, 0.5, 3.0, with_diagonal=bool(rng.integers(0, 2)))
    conn = net["connections"]
    conn[...] &= (rng.random(conn.shape) < float(rng.choice([0.3, 0.8, 1.0])))
    lat = net["lattice"].astype(np.int64)
    cross = lat[:, None] != lat[None, :]
    sym = conn[:nn] | conn[:nn].T
    conn[:nn] = np.where(cross, sym, conn[:nn])
    one_way = cross & (role[lat] == FIXED)[:, None] & (rng.random((nn, nn)) < 0.2)        # extra p -> q, p in a fixed lattice
    conn[:nn] |= one_way.astype(conn.dtype)
    net["weights"][...] = ob.uniform_array(seed + 7, conn.size, 0.5, 3.0).reshape(conn.shape) * conn
    for a in range(n_lat):
        for b in range(a + 1, n_lat):
            pair = tuple(sorted((int(role[a]), int(role[b]))))
            net["conn_kind"][a, b] = net["conn_kind"][b, a] = int(rng.choice(ALLOWED[pair]))
    for s in range(len(st)):
        for b in range(n_lat):
            net["conn_kind"][n_lat + s, b] = int(rng.choice(FROM_CELLS[int(role[b])]))
    if violation == 1:                  # a connection of a visited lattice without its reverse
        net["conn_kind"][0, 1] = net["conn_kind"][1, 0] = 1
        first = net.layout.ranges()
        p0, q0 = first[lattices[0][0]][0], first[lattices[1][0]][0]
        conn[p0, q0], conn[q0, p0] = 1, 0
        net["weights"][p0, q0], net["weights"][q0, p0] = 1.0, 0.0
    if violation == 2:                  # reward-modulated weights, no modulator on either side, one side plastic
        net["conn_kind"][0, 1] = net["conn_kind"][1, 0] = 1
    if violation == 3:                  # plain weights between a plastic plain lattice and a modulated one
        net["conn_kind"][0, 1] = net["conn_kind"][1, 0] = 2
    if violation in (2, 3):
        first = net.layout.ranges()
        p0, q0 = first[lattices[0][0]][0], first[lattices[1][0]][0]
        conn[p0, q0] = conn[q0, p0] = 1
        net["weights"][p0, q0] = net["weights"][q0, p0] = 1.0
    exists = conn != 0
    net["traces"][...] = ob.uniform_array(seed + 9, conn.size, -0.001, 0.001).reshape(conn.shape) * exists
    net["pending"][...] = ob.uniform_array(seed + 10, conn.size, -0.01, 0.01).reshape(conn.shape) * exists * (rng.random(conn.shape) < 0.3)
    net["edge_counter"][...] = (rng.random(conn.shape) < 0.5) & exists
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    for k in ("dt", "st_dt", "stdp_dt", "rm_dt"):
        net[k] = dt
    if paused:
        slot = int(np.flatnonzero(role == MOD)[seed % int((role == MOD).sum())])
        net["rm_do_modulation"][slot] = 0
        net["rm_is_modulated"][slot] = 1
    steps = int(rng.integers(60, 200))
    rewards = ob.uniform_array(seed + 11, steps, -0.02, 0.03)
    rewards[::3] = 0.0
    return net, steps, rewards
