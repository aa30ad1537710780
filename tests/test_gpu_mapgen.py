"""Device map generation / pool refresh: `mg_set_generated_maps`, `mg_refresh_maps`, `mg_get_map`.

The reference builds every episode's map in Python (`simulator.py:264-267`); its random builders are a border plus one
uniform shuffle (`map_builder/random_map.py:29-96`, `mapgen/scenes/random.py:19-69`).  Numpy's bit stream is not part
of parity for maps (they are inputs to the step path), the distribution is.  These tests pin:
  * structure: border, rooms, per-room symbol counts -- equal to the 64 maps the reference's MapGen produced
    (tests/golden/arena24_map_pool.npz);
  * uniformity: chi-square of the per-cell occupancy over thousands of generated maps, for both placement algorithms
    (sparse rooms: rejection sampling; dense rooms: in-place shuffle);
  * that what the device derives from a generated map (object ids, spawn / assembler / chest order, episode-start
    templates) is what the host derives from the same bytes: the environments step bit-exactly like the oracle
    built on the read-back map, through auto-resets and pool refreshes;
  * a refresh never changes a map under a running episode, and gives every episode a new map.
"""

import numpy as np
import pytest

from metta_amd.builder import make_arena, make_combat_stress
from metta_amd.compile import compile_game
from metta_amd.mapgen import DeviceMapSpec, GeneratedMaps, RandomMapBuilder
from tests.helpers import digest, load_map_pool
from tests.test_gpu_parity import _compare_state, _compare_step

pytestmark = pytest.mark.gpu


def _gen_batch(cfg, num_envs, spec, num_maps, seed, seeds=None, map_ids=None):
    from metta_amd.batch import MettaGridBatch

    return MettaGridBatch(cfg, num_envs, GeneratedMaps(spec, num_maps, seed), device=0, seeds=seeds, map_ids=map_ids)


def _chi2(counts: np.ndarray) -> float:
    exp = counts.sum() / counts.size
    return float(((counts - exp) ** 2 / exp).sum())


def test_arena_pool_has_the_reference_structure_and_is_uniform():
    cfg = make_arena(24)
    comp = compile_game(cfg, init_obs_passes=2)
    spec = cfg.game.map_builder.device_spec()
    assert (spec.height, spec.width, len(spec.rooms)) == (62, 62, 4)
    P = 2048
    b = _gen_batch(cfg, 2, spec, P, seed=5)
    maps = np.stack([b.get_map(m) for m in range(P)])
    ref = np.stack([comp.encode_map(m) for m in load_map_pool()])  # the reference's MapGen output, same cell codes
    wall = comp.cell_code["wall"]
    outside = np.ones((62, 62), bool)
    for r0, c0, h, w in spec.rooms:
        outside[r0 : r0 + h, c0 : c0 + w] = False
    assert (maps[:, outside] == wall).all() and (ref[:, outside] == wall).all()
    ncodes = len(comp.cell_names) + 1
    want = np.zeros(ncodes, np.int64)
    for name, n in spec.room_symbols[0]:
        want[comp.cell_code[name]] += n
    want[0] = 625 - want.sum()
    occ = np.zeros((ncodes, 625), np.int64)
    for r0, c0, h, w in spec.rooms:
        room = maps[:, r0 : r0 + h, c0 : c0 + w].reshape(P, -1)
        counts = np.stack([(room == k).sum(1) for k in range(ncodes)], 1)
        assert (counts == want).all(), "every room holds exactly the configured symbols"
        rroom = ref[:, r0 : r0 + h, c0 : c0 + w].reshape(len(ref), -1)
        assert (np.stack([(rroom == k).sum(1) for k in range(ncodes)], 1) == want).all(), "and so does the reference's"
        for k in range(ncodes):
            occ[k] += (room == k).sum(0)
    # uniform placement: per kind, occupancy of the 625 room cells over 4 * 2048 rooms (df = 624: mean 624, sd 35)
    for k in range(1, ncodes):
        if want[k]:
            assert _chi2(occ[k]) < 624 + 6 * 35.4, f"{comp.cell_names[k - 1]}: chi2 {_chi2(occ[k]):.0f}"
    # two symbols of a room are (almost) independent: P(cell i is an agent and its right neighbour too) = 6*5/(625*624)
    ag = comp.cell_code["agent.agent"]
    pairs = 0
    for r0, c0, h, w in spec.rooms:
        room = maps[:, r0 : r0 + h, c0 : c0 + w] == ag
        pairs += int((room[:, :, :-1] & room[:, :, 1:]).sum())
    exp_pairs = 4 * P * 25 * 24 * 30 / (625 * 624)
    assert abs(pairs - exp_pairs) < 6 * np.sqrt(exp_pairs), (pairs, exp_pairs)
    assert len({digest(m) for m in maps}) == P, "all slots differ"
    # slot m is a function of (seed, generation, m): same seed -> same pool, other seed -> another
    b2 = _gen_batch(cfg, 2, spec, 8, seed=5)
    b3 = _gen_batch(cfg, 2, spec, 8, seed=6)
    for m in range(8):
        assert np.array_equal(b2.get_map(m), maps[m])
        assert not np.array_equal(b3.get_map(m), maps[m])
    for x in (b, b2, b3):
        x.close()


def test_dense_room_uses_the_shuffle_and_is_uniform():
    """More symbols than half the room: the generator shuffles the room in place (numpy's own algorithm)."""
    from metta_amd.config import MettaGridConfig

    cfg = MettaGridConfig.EmptyRoom(num_agents=4, width=10, height=10, border_width=1)
    mb = RandomMapBuilder(agents=4, width=10, height=10, border_width=1, objects={"wall": 40})
    spec = mb.device_spec()
    comp = compile_game(cfg, init_obs_passes=2)
    P = 4096
    b = _gen_batch(cfg, 1, spec, P, seed=1)
    maps = np.stack([b.get_map(m) for m in range(P)])
    wall, ag = comp.cell_code["wall"], comp.cell_code["agent.agent"]
    assert (maps[:, 0, :] == wall).all() and (maps[:, -1, :] == wall).all() and (maps[:, :, 0] == wall).all()
    inner = maps[:, 1:-1, 1:-1].reshape(P, 64)
    assert ((inner == wall).sum(1) == 40).all() and ((inner == ag).sum(1) == 4).all()
    for k in (wall, ag, 0):
        assert _chi2((inner == k).sum(0)) < 63 + 6 * 11.3  # df = 63
    # the reference's builder halves object counts that do not fit (random_map.py:62-71): so does the spec
    big = RandomMapBuilder(agents=4, width=6, height=6, border_width=1, objects={"wall": 40}).device_spec()
    assert dict(big.room_symbols[0])["wall"] == 10
    b.close()


@pytest.mark.parametrize("path_env", [{}, {"MG_FORCE_GENERAL": "1"}], ids=["fused", "general"])
def test_refresh_gives_every_episode_a_new_map_and_never_changes_one_in_use(path_env, monkeypatch):
    """2E slots, stride E: an environment alternates between two slots and the idle one is redrawn.  The oracle is
    rebuilt on the map read back at each episode start; bit-exact parity at every step proves that the derived
    lists, object ids and episode-start templates match the bytes, and that no map changed under an episode."""
    from oracle.mg_oracle import make_oracle_env

    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    cfg = make_arena(24)
    cfg.game.max_steps = 25
    comp = compile_game(cfg, init_obs_passes=2)
    spec = cfg.game.map_builder.device_spec()
    E, A = 3, 24
    seeds, cur_map = [21, 22, 23], [0, 1, 2]
    b = _gen_batch(cfg, E, spec, 2 * E, seed=9, seeds=seeds, map_ids=cur_map)
    b.set_auto_reset(True, E)
    oras = [make_oracle_env(comp, b.get_map(m), s) for s, m in zip(seeds, cur_map)]
    _compare_step(b, oras, 0, A)
    played = [[digest(b.get_map(m))] for m in cur_map]
    rs = np.random.RandomState(3)
    for t in range(1, 160):
        if t % 10 == 0:
            before = [b.get_map(m) for m in range(2 * E)]
            b.refresh_maps()
            after = [b.get_map(m) for m in range(2 * E)]
            for e in range(E):
                assert np.array_equal(before[cur_map[e]], after[cur_map[e]]), "a map in use was redrawn"
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + E) % (2 * E)
                m = b.get_map(cur_map[e])
                played[e].append(digest(m))
                oras[e] = make_oracle_env(comp, m, seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
        if t % 40 == 0:
            _compare_state(b, oras, t)
    for e in range(E):
        assert len(played[e]) == 7 and len(set(played[e])) == 7, "every episode of an environment had its own map"
    # an unused, freshly drawn slot is left alone; a forced refresh redraws everything and asks for a reset
    b.refresh_maps()
    snap = [b.get_map(m) for m in range(2 * E)]
    b.refresh_maps()
    assert all(np.array_equal(x, b.get_map(m)) for m, x in enumerate(snap))
    b.refresh_maps(force=True)
    assert not any(np.array_equal(x, b.get_map(m)) for m, x in enumerate(snap))
    with pytest.raises(RuntimeError, match="mg_reset"):
        b.step(np.zeros(E * A, np.int32))
    b.check_errors()
    b.close()


def test_generated_combat_map_in_hbm_grid_mode_vs_oracle():
    """256 x 256, 1024 agents in teams (RandomMapBuilder): the kernels keep grids in HBM; template rebuilds of a refresh
    use their own grid area, not the running environments'."""
    from oracle.mg_oracle import make_oracle_env

    cfg = make_combat_stress(1024, 256)
    cfg.game.max_steps = 20
    comp = compile_game(cfg, init_obs_passes=2)
    spec = cfg.game.map_builder.device_spec()
    E, A = 2, 1024
    seeds, cur_map = [900, 901], [0, 1]
    b = _gen_batch(cfg, E, spec, 2 * E, seed=4, seeds=seeds, map_ids=cur_map)
    b.set_auto_reset(True, E)
    m0 = b.get_map(0)
    want = {comp.cell_code[k]: n for k, n in spec.room_symbols[0]}
    assert all(int((m0 == c).sum()) == n + (2 * 256 + 2 * 254 if comp.cell_names[c - 1] == "wall" else 0) for c, n in want.items())
    oras = [make_oracle_env(comp, b.get_map(m), s) for s, m in zip(seeds, cur_map)]
    nA = len(comp.action_names)
    rs = np.random.RandomState(2)
    for t in range(1, 50):
        if t % 7 == 0:
            b.refresh_maps()
        acts = np.where(rs.rand(E, A) < 0.7, rs.randint(1, 5, size=(E, A)), rs.randint(0, nA, size=(E, A))).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + E) % (2 * E)
                oras[e] = make_oracle_env(comp, b.get_map(cur_map[e]), seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 49)
    b.check_errors()
    b.close()


def test_generator_rejects_bad_specs():
    from metta_amd.batch import MettaGridBatch

    cfg = make_arena(24)
    good = cfg.game.map_builder.device_spec()
    b = MettaGridBatch(cfg, 1, load_map_pool()[:2], device=0)
    with pytest.raises(RuntimeError, match="not made by mg_set_generated_maps"):
        b.refresh_maps()
    few = DeviceMapSpec(62, 62, "wall", good.rooms[:3], good.room_symbols[:3])
    with pytest.raises(RuntimeError, match="18 agents but the config says num_agents = 24"):
        b.set_generated_maps(few, 4)
    overlap = DeviceMapSpec(62, 62, "wall", [good.rooms[0]] * 4, good.room_symbols)
    with pytest.raises(RuntimeError, match="overlap"):
        b.set_generated_maps(overlap, 4)
    outside = DeviceMapSpec(62, 62, "wall", [(50, 50, 25, 25)] + good.rooms[1:], good.room_symbols)
    with pytest.raises(RuntimeError, match="outside the map"):
        b.set_generated_maps(outside, 4)
    crowded = DeviceMapSpec(62, 62, "wall", good.rooms, [[("wall", 700)] + s for s in good.room_symbols])
    with pytest.raises(RuntimeError, match="symbols but only 625 cells"):
        b.set_generated_maps(crowded, 4)
    with pytest.raises(RuntimeError, match="Unknown object type"):
        b.set_generated_maps(DeviceMapSpec(62, 62, "wall", good.rooms, [[("altar", 1)] + s for s in good.room_symbols]), 4)
    # the handle survives: a good spec works afterwards
    b.set_generated_maps(good, 4, seed=1)
    b.reset()
    b.step(np.zeros(24, np.int32))
    b.check_errors()
    b.close()


def test_vecenv_with_device_maps_refreshes_without_host_syncs():
    """`MettaGridDeviceVecEnv(device_maps=True)`: 2E slots, stride E, a stream-ordered refresh every max_steps / 2
    steps; send() never blocks on the device and every episode of an environment sees a new map."""
    import torch

    from metta_amd.vecenv import MettaGridDeviceVecEnv

    cfg = make_arena(24)
    cfg.game.max_steps = 20
    E = 8
    env = MettaGridDeviceVecEnv(cfg, E, device=0, seed=3, early_reset=False, device_maps=True, map_seed=11)
    assert env._map_refresh_every == 10 and len(env.batch.map_pool) == 2 * E and env.batch.map_stride == E
    env.async_reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    seen = [set() for _ in range(E)]
    infos = []
    for t in range(1, 106):
        o, r, d, tr, ta, info, ids, mask = env.recv()
        infos += info
        env.send(torch.randint(0, 5, (E * 24,), generator=g, device="cuda", dtype=torch.int32))
        if t % 21 == 10:  # mid-episode (episodes restart every 21 sends): which slot, which map
            table = env.batch.env_table()
            for e in range(E):
                assert int(table[e, 3]) % E == e
                seen[e].add(digest(env.batch.get_map(int(table[e, 3]))))
    assert env.host_syncs <= 5, "host reads only to collect the five rounds of finished episodes, none for the maps"
    env.flush_infos()
    infos += env.infos
    assert len(infos) >= 4 * E
    for e in range(E):
        assert len(seen[e]) == 5, "five episodes, five maps"
    env.batch.check_errors()
    env.close()


@pytest.mark.parametrize("path_env", [{}, {"MG_FORCE_GENERAL": "1"}], ids=["fused", "general"])
def test_host_built_maps_written_into_idle_slots_vs_oracle(path_env, monkeypatch):
    """`mg_write_map_slot`: the reference's per-episode `map_builder.build()` (simulator.py:264-267) for maps the host
    builds.  An uploaded pool of 2E slots, stride E; before every episode the idle twin slot receives another of the
    reference's 64 MapGen maps.  Bit-exact parity with the oracle built on that map proves that the slot's derived
    entries (object ids, spawn / assembler / chest lists, counts) and its episode-start template match the bytes."""
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    cfg = make_arena(24)
    cfg.game.max_steps = 20
    comp = compile_game(cfg, init_obs_passes=2)
    ref = [comp.encode_map(m) for m in load_map_pool()]
    E, A = 3, 24
    seeds, cur_map = [31, 32, 33], [0, 1, 2]
    b = MettaGridBatch(cfg, E, ref[: 2 * E], device=0, seeds=seeds, map_ids=cur_map)
    b.set_auto_reset(True, E)
    slot_map = {m: ref[m] for m in range(2 * E)}
    nxt = 2 * E
    oras = [make_oracle_env(comp, slot_map[m], s) for s, m in zip(seeds, cur_map)]
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(8)
    for t in range(1, 110):
        if t % 20 == 10:  # mid-episode: fill every idle slot with a map not played yet
            for e in range(E):
                idle = (cur_map[e] + E) % (2 * E)
                b.write_map_slot(idle, ref[nxt % len(ref)])
                slot_map[idle] = ref[nxt % len(ref)]
                assert np.array_equal(b.map_pool[idle], slot_map[idle])
                nxt += 1
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + E) % (2 * E)
                oras[e] = make_oracle_env(comp, slot_map[cur_map[e]], seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
        if t % 35 == 0:
            _compare_state(b, oras, t)
    assert nxt == 2 * E + 5 * E
    b.check_errors()
    b.close()


def test_write_map_slot_on_a_generated_pool_and_its_errors():
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    cfg = make_arena(24)
    cfg.game.max_steps = 12
    comp = compile_game(cfg, init_obs_passes=2)
    spec = cfg.game.map_builder.device_spec()
    ref = [comp.encode_map(m) for m in load_map_pool()]
    b = _gen_batch(cfg, 1, spec, 2, seed=2, seeds=[5], map_ids=[0])
    b.set_auto_reset(True, 1)
    b.write_map_slot(1, ref[7])  # the slot the environment enters next
    assert np.array_equal(b.get_map(1), ref[7])
    # a written slot counts as fresh: a refresh leaves it alone until an episode has used it
    b.refresh_maps()
    assert np.array_equal(b.get_map(1), ref[7])
    ora = make_oracle_env(comp, b.get_map(0), 5)
    rs = np.random.RandomState(1)
    cur = 0
    for t in range(1, 30):
        acts = rs.randint(0, 5, size=(1, 24)).astype(np.int32)
        if ora.terminals.all() or ora.truncations.all():
            cur ^= 1
            ora = make_oracle_env(comp, b.get_map(cur), 5)
        ora.step(acts[0])
        b.step(acts.reshape(-1))
        _compare_step(b, [ora], t, 24)
    bad = ref[3].copy()
    bad[np.isin(bad, [c + 1 for c, k in enumerate(comp.cell_kinds) if k == 2])] = 0
    with pytest.raises(RuntimeError, match="holds 0 agents but the config says num_agents = 24"):
        b.write_map_slot(0, bad)
    crowded = ref[3].copy()
    crowded[crowded == 0] = [c + 1 for c, k in enumerate(comp.cell_kinds) if k == 3][0]
    with pytest.raises(RuntimeError, match="the pool was sized for"):
        b.write_map_slot(0, crowded)
    with pytest.raises(RuntimeError, match="slot out of range"):
        b.write_map_slot(2, ref[3])
    with pytest.raises(RuntimeError, match="the pool's slots are"):
        b.write_map_slot(0, ref[3][:40])
    unknown = ref[3].copy()
    unknown[1, 1] = 200
    with pytest.raises(RuntimeError, match="Unknown object type"):
        b.write_map_slot(0, unknown)
    b.check_errors()
    b.close()
    # pools of mixed sizes have no slot to write to
    small = np.zeros((20, 20), np.uint8)
    small[2:8, 2:6] = comp.cell_code["agent.agent"]
    b2 = MettaGridBatch(cfg, 1, [ref[0], small], device=0)
    with pytest.raises(RuntimeError, match="differ in size"):
        b2.write_map_slot(0, ref[1])
    b2.close()


def test_clipper_on_a_refreshed_device_pool_vs_oracle():
    """A Clipper over device-generated maps: every refresh rebuilds the slot's template with the template's own clipper,
    every restart rebuilds the environment's (stations that start clipped draw their unclipping protocol from the
    environment's seed; weights and length scale follow the new map)."""
    from metta_amd.config import ClipperConfig, ProtocolConfig
    from oracle.mg_oracle import make_oracle_env

    cfg = make_arena(24)
    cfg.game.max_steps = 20
    cfg.game.clipper = ClipperConfig(
        unclipping_protocols=[ProtocolConfig(input_resources={"ore_red": 1}, cooldown=2), ProtocolConfig(input_resources={"battery_red": 1}, cooldown=1)],
        clip_period=2, scaled_cutoff_distance=2,
    )  # fmt: skip
    cfg.game.objects["generator_red"].start_clipped = True
    cfg.game.objects["lasery"].clip_immune = True
    comp = compile_game(cfg, init_obs_passes=2)
    spec = cfg.game.map_builder.device_spec()
    E, A = 3, 24
    seeds, cur_map = [61, 62, 63], [0, 1, 2]
    b = _gen_batch(cfg, E, spec, 2 * E, seed=12, seeds=seeds, map_ids=cur_map)
    b.set_auto_reset(True, E)
    oras = [make_oracle_env(comp, b.get_map(m), s) for s, m in zip(seeds, cur_map)]
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(4)
    for t in range(1, 110):
        if t % 10 == 0:
            b.refresh_maps()
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + E) % (2 * E)
                oras[e] = make_oracle_env(comp, b.get_map(cur_map[e]), seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
        if t % 15 == 0:
            _compare_state(b, oras, t)
            for e, o in enumerate(oras):
                ls, w = o.clipper_state()
                dls, dw = b.clipper_state(e)
                assert dls == ls and np.array_equal(dw, w), f"step {t} env {e}: clipper state"
    st = b.assembler_state(0)
    assert (st[:, 4] >= 2).any(), "stations were clipped"
    b.check_errors()
    b.close()
