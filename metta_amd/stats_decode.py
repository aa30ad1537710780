"""Raw stat slots -> the reference's `get_episode_stats()` dictionaries
(`BIND/mettagrid_c.cpp:936-953`: {"game": {name: float}, "agent": [{name: float}, ...]}); a key is present only
if its slot was touched."""

from __future__ import annotations

import numpy as np


def decode_episode_stats(compiled, agent_values, agent_touched, game_values, game_touched, agent_groups) -> dict:
    game = {
        compiled.game_slot_names[i]: float(game_values[i]) for i in np.nonzero(np.asarray(game_touched))[0].tolist()
    }
    agents = []
    for a in range(np.asarray(agent_values).shape[0]):
        names = compiled.agent_slot_names[agent_groups[a]]
        agents.append({names[i]: float(agent_values[a][i]) for i in np.nonzero(np.asarray(agent_touched[a]))[0].tolist()})
    return {"game": game, "agent": agents}
