"""Raw stat slots -> the reference's `get_episode_stats()` dictionaries
(`BIND/mettagrid_c.cpp:936-953`: {"game": {name: float}, "agent": [{name: float}, ...]}); a key is present only
if its slot was touched."""

from __future__ import annotations

import numpy as np

from metta_amd.stats import invalid_table_keys


def decode_episode_stats(compiled, agent_values, agent_touched, game_values, game_touched, agent_groups) -> dict:
    game = {
        compiled.game_slot_names[i]: float(game_values[i]) for i in np.nonzero(np.asarray(game_touched))[0].tolist()
    }
    agents = []
    for a in range(np.asarray(agent_values).shape[0]):
        names = compiled.agent_slot_names[agent_groups[a]]
        d = {
            names[i]: float(agent_values[a][i])
            for i in np.nonzero(np.asarray(agent_touched[a]))[0].tolist()
            if names[i] and not names[i].startswith("\x00")
        }
        d.update(invalid_table_keys(agent_values[a]))  # action.invalid_index.<id> for ids without a slot of their own
        agents.append(d)
    return {"game": game, "agent": agents}
