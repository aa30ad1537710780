"""Stat slot <-> name tables (host-side mirror of `MgAgentSlots` / `MgGameSlots` in csrc/mg_blob.h).

The reference keys its per-agent and per-game statistics by strings built at run time
(`CPP/systems/stats_tracker.hpp:35-46`; key list in SURVEY.md §8a-S).  The device keeps fixed f32 slots plus a
"touched" bit per slot; this module maps slots back to the reference's exact key strings so that
`get_episode_stats()` returns the same dictionaries (a key exists only once touched).
"""

from __future__ import annotations

HANDLER_NAMES = ["noop", "move", "change_vibe"]
INVALID_VARIANTS = 9
INVALID_TABLE = 4  # MG_INVALID_TABLE: (id, count) slot pairs at the end of an agent's row for ids without a slot of their own
INVALID_TABLE_MARK = "\x00action.invalid_index.#"  # name of the k-th count slot (never a key: see invalid_table_keys)


def agent_slot_names(resource_names: list[str], group_names: list[str], own_group: str, n_actions: int) -> list[str]:
    R, G = len(resource_names), len(group_names)
    names: list[str] = [""] * (26 + 7 * R + 3 * G + G * R + 2 * INVALID_TABLE)
    for k in range(INVALID_TABLE):  # (the id slots stay unnamed)
        names[26 + 7 * R + 3 * G + G * R + 2 * k + 1] = f"{INVALID_TABLE_MARK}{k}"
    for h, hn in enumerate(HANDLER_NAMES):
        names[h] = f"action.{hn}.success"
        names[3 + h] = f"action.{hn}.failed"
    names[6] = "action.failed"
    names[7] = "action.invalid_index"
    names[8] = "action.invalid_index.-1"
    for v in range(1, 8):
        names[8 + v] = f"action.invalid_index.{n_actions + v - 1}"
    names[16] = "action.invalid_index.other"
    names[17] = "actions.swap"
    names[18] = "status.frozen.ticks"
    names[19] = f"status.frozen.ticks.{own_group}"
    names[20] = "status.max_steps_without_motion"
    names[21] = "inventory.diversity"
    names[22] = "damage.triggered"
    names[23] = "damage.items_lost"
    names[24] = f"action.transfer.{own_group}.count"
    names[25] = f"action.attack.{own_group}.friendly_fire"
    b = 26
    for r, rn in enumerate(resource_names):
        names[b + 0 * R + r] = f"{rn}.gained"
        names[b + 1 * R + r] = f"{rn}.lost"
        names[b + 2 * R + r] = f"{rn}.amount"
        names[b + 3 * R + r] = f"damaged.{rn}"
        names[b + 4 * R + r] = f"chest.{rn}.deposited_by_agent"
        names[b + 5 * R + r] = f"chest.{rn}.withdrawn_by_agent"
        names[b + 6 * R + r] = f"action.transfer.{own_group}.{rn}.self"
    b += 7 * R
    for g, gn in enumerate(group_names):
        names[b + 0 * G + g] = f"action.attack.{own_group}.hit.{gn}"
        names[b + 1 * G + g] = f"action.attack.{own_group}.hit_by.{gn}"
        names[b + 2 * G + g] = f"action.attack.{own_group}.blocked_by.{gn}"
        for r, rn in enumerate(resource_names):
            names[b + 3 * G + g * R + r] = f"action.transfer.{own_group}.{rn}.to.{gn}"
    return names


def game_slot_names(resource_names: list[str], cell_names: list[str]) -> list[str]:
    """cell_names[code-1] = the map cell name of cell code `code`."""
    R, C = len(resource_names), len(cell_names)
    names = ["tokens_written", "tokens_dropped", "tokens_free_space"]
    names += [f"objects.{c}" for c in cell_names]
    names += [f"assembler.{r}.created" for r in resource_names]
    names += [f"chest.{r}.deposited" for r in resource_names]
    names += [f"chest.{r}.withdrawn" for r in resource_names]
    names += [f"chest.{r}.amount" for r in resource_names]
    assert len(names) == 3 + C + 4 * R
    return names


def invalid_table_keys(values_row) -> dict[str, float]:
    """The reference's `action.invalid_index.<id>` keys (mettagrid_c.cpp:565-570) held in the (id, count) pairs at the end
    of one agent's slot row: ids are raw int32 bits in the f32 slots."""
    import numpy as np

    row = np.ascontiguousarray(values_row, dtype=np.float32)
    n = row.shape[0]
    out: dict[str, float] = {}
    for k in range(INVALID_TABLE):
        cnt = float(row[n - 2 * INVALID_TABLE + 2 * k + 1])
        if cnt != 0.0:
            ident = int(row[n - 2 * INVALID_TABLE + 2 * k : n - 2 * INVALID_TABLE + 2 * k + 1].view(np.int32)[0])
            out[f"action.invalid_index.{ident}"] = cnt
    return out
