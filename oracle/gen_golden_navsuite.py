"""Navigation evaluation suite fixture (BASELINE configs[2]; run in the build container only).

    make -C oracle ref && python oracle/gen_golden_navsuite.py

For every entry of the reference's `NAVIGATION_EVALS` (recipes/experiment/cfg.py) builds the environment the way
`make_nav_ascii_env` does (recipes/experiment/navigation.py:32-59): the ASCII map
packages/mettagrid/configs/maps/navigation_sequence/<name>.map with `n` / `m` remapped to `assembler`, wrapped by the
reference's `MapGen(instances=4, border_width=6, instance_border_width=3)`, `make_navigation(4)` with the entry's
`max_steps` and the 0.333 heart reward -- all with the reference's own Python (map builders, config classes, the real
`convert_to_cpp_game_config`) and its compiled core.  Records the generated maps (a ragged pool: sizes differ), the
per-map `max_steps`, a seeded action stream and, per map, the observation / reward digests of a whole episode plus
final stats.  Nothing of the reference is stored but data: maps as cell codes over a small name table.
"""

from __future__ import annotations

import json
import runpy
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle.gen_golden import GOLD, digest, install_reference_shims  # noqa: E402

REF = Path("/root/reference")


def main() -> None:
    install_reference_shims()
    import mettagrid.builder.envs as eb
    from mettagrid.config.mettagrid_c_config import convert_to_cpp_game_config
    from mettagrid.config.mettagrid_config import AsciiMapBuilder
    from mettagrid.mapgen.mapgen import MapGen
    from mettagrid.mettagrid_c import MettaGrid

    evals = runpy.run_path(str(REF / "recipes/experiment/cfg.py"))["NAVIGATION_EVALS"]
    names_all = ["agent.agent", "assembler", "empty", "wall"]
    idx = {n: i for i, n in enumerate(names_all)}
    maps, dims, max_steps, entries = [], [], [], []
    digests = {}
    for k, ev in enumerate(evals):
        name, ms, n_inst = ev["name"], int(ev["max_steps"]), int(ev["num_instances"])
        env = eb.make_navigation(num_agents=ev["num_agents"] * n_inst)
        env.game.max_steps = ms
        inst = AsciiMapBuilder.Config.from_uri(str(REF / f"packages/mettagrid/configs/maps/navigation_sequence/{name}.map"))
        inst.char_to_map_name["n"] = "assembler"
        inst.char_to_map_name["m"] = "assembler"
        env.game.map_builder = MapGen.Config(instances=n_inst, border_width=6, instance_border_width=3, instance=inst, seed=1000 + k)
        env.game.agent.rewards.inventory["heart"] = 0.333
        grid = env.game.map_builder.create().build().grid
        uniq = set(str(x) for x in np.unique(grid))
        assert uniq <= set(names_all), (name, uniq)
        codes = np.vectorize(lambda s: idx[str(s)], otypes=[np.uint8])(grid)
        A = env.game.num_agents
        assert int((codes == idx["agent.agent"]).sum()) == A == 4, name
        # one whole episode on the reference core with the real converter
        c_cfg = convert_to_cpp_game_config(env.game.model_dump())
        seed = 7000 + k
        core = MettaGrid(c_cfg, grid.tolist(), seed)
        T = env.game.obs.num_tokens
        obs = np.zeros((A, T, 3), np.uint8)
        term, trunc = np.zeros(A, bool), np.zeros(A, bool)
        rew, act = np.zeros(A, np.float32), np.zeros(A, np.int32)
        core.set_buffers(obs, term, trunc, rew, act)
        obs0 = digest(obs)
        rs = np.random.RandomState(seed)
        acts = rs.randint(0, 5, size=(ms, A)).astype(np.int32)
        od = np.zeros(ms, np.uint64)
        rsum = 0.0
        for t in range(ms):
            act[:] = acts[t]
            core.step()
            od[t] = digest(obs)
            rsum += float(rew.sum())
        assert term.all() or trunc.all(), name
        maps.append(codes.reshape(-1))
        dims.append(codes.shape)
        max_steps.append(ms)
        entries.append({"name": name, "seed": seed, "obs0_digest": obs0, "obs_final_digest": int(od[-1]),
                        "obs_mid_digest": int(od[ms // 2]), "reward_sum": rsum, "stats": core.get_episode_stats(),
                        "episode_rewards": [float(x) for x in core.get_episode_rewards()],
                        "terminated": bool(term.all()), "truncated": bool(trunc.all())})
        digests[name] = od
        print(f"{name}: {codes.shape}, max_steps {ms}, reward sum {rsum:.3f}")
    meta = {"cell_names": names_all, "entries": entries, "heart_reward": 0.333,
            "action_spec": "RandomState(seed).randint(0, 5, size=(max_steps, 4))"}
    np.savez_compressed(
        GOLD / "navigation_suite.npz",
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        maps=np.concatenate(maps), dims=np.array(dims, np.int32), max_steps=np.array(max_steps, np.int32),
        **{f"od_{n}": d for n, d in digests.items()},
    )
    print("wrote navigation_suite.npz:", len(entries), "maps,", (GOLD / "navigation_suite.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
