"""Host-side game configuration for the MI355X MettaGrid path.

Same names, fields, defaults and validation behaviour as the reference's pydantic configuration
(`packages/mettagrid/python/src/mettagrid/config/mettagrid_config.py:35-613`, `obs_config.py:12-25`,
`vibes.py`), so that a config written for the reference constructs unchanged here.  Only what the
environment-step path consumes is kept: map builders are replaced by `metta_amd.mapgen` (maps are inputs
to the path) and rendering fields are carried but unused.
"""

from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Literal, Optional, Union, get_args

from pydantic import BaseModel, ConfigDict, Field, model_validator
from typing_extensions import Annotated


class Config(BaseModel):
    """Strict pydantic base (reference: `PY/base_config.py` — extra fields are errors)."""

    model_config = ConfigDict(extra="forbid", validate_assignment=True, populate_by_name=True)


# --------------------------------------------------------------------------------------------------------
# Vibes (reference: PY/config/vibes.py — 152 named vibes; only the names are part of the step-path contract:
# they name the change_vibe_<name> actions and decide `Attack::_is_vibing_resource`, attack.hpp:179-186).
# --------------------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class Vibe:
    symbol: str
    name: str
    category: str = "misc"


def _load_vibe_names() -> list[str]:
    path = Path(__file__).parent / "data" / "vibe_names.txt"
    return [ln.strip() for ln in path.read_text().splitlines() if ln.strip()]


VIBES: list[Vibe] = [Vibe("", n) for n in _load_vibe_names()]
VIBE_BY_NAME: dict[str, Vibe] = {v.name: v for v in VIBES}


@dataclass(frozen=True)
class Action:
    """reference: PY/simulator/types.py — an action is identified by its name."""

    name: str


Direction = Literal["north", "south", "east", "west", "northeast", "northwest", "southeast", "southwest"]
Directions = list(get_args(Direction))
CardinalDirection = Literal["north", "south", "west", "east"]
CardinalDirections = list(get_args(CardinalDirection))


class ObsConfig(Config):
    width: int = Field(default=11)
    height: int = Field(default=11)
    token_dim: int = Field(default=3)
    num_tokens: int = Field(default=200)
    token_value_base: int = Field(default=256)


class AgentRewards(Config):
    inventory: dict[str, float] = Field(default_factory=dict)
    inventory_max: dict[str, float] = Field(default_factory=dict)
    stats: dict[str, float] = Field(default_factory=dict)
    stats_max: dict[str, float] = Field(default_factory=dict)


class ResourceLimitsConfig(Config):
    limit: int
    resources: list[str]
    modifiers: dict[str, int] = Field(default_factory=dict)


class InventoryConfig(Config):
    default_limit: int = Field(default=65535, ge=0)
    limits: dict[str, ResourceLimitsConfig] = Field(default_factory=dict)
    initial: dict[str, int] = Field(default_factory=dict)
    regen_amounts: dict[str, dict[str, int]] = Field(default_factory=dict)

    def get_limit(self, resource_name: str) -> int:
        for limit_config in self.limits.values():
            if resource_name in limit_config.resources:
                return limit_config.limit
        return self.default_limit


class DamageConfig(Config):
    threshold: dict[str, int] = Field(default_factory=dict)
    resources: dict[str, int] = Field(default_factory=dict)

    @model_validator(mode="after")
    def _validate_distinct_keys(self) -> "DamageConfig":
        overlap = set(self.threshold) & set(self.resources)
        if overlap:
            raise ValueError(
                "Resources cannot appear in both threshold and resources maps. "
                f"Overlapping keys: {sorted(overlap)}"
            )
        return self


class AgentConfig(Config):
    inventory: InventoryConfig = Field(default_factory=InventoryConfig)
    rewards: AgentRewards = Field(default_factory=AgentRewards)
    freeze_duration: int = Field(default=10, ge=-1)
    team_id: int = Field(default=0, ge=0)
    tags: list[str] = Field(default_factory=lambda: ["agent"])
    diversity_tracked_resources: list[str] = Field(default_factory=list)
    initial_vibe: int = Field(default=0, ge=0)
    damage: Optional[DamageConfig] = Field(default=None)


class ActionConfig(Config):
    action_handler: str
    enabled: bool = Field(default=True)
    required_resources: dict[str, int] = Field(default_factory=dict)
    consumed_resources: dict[str, int] = Field(default_factory=dict)

    def actions(self) -> list[Action]:
        return self._actions() if self.enabled else []

    def _actions(self) -> list[Action]:  # pragma: no cover - abstract
        raise NotImplementedError


class NoopActionConfig(ActionConfig):
    action_handler: str = Field(default="noop")

    def _actions(self) -> list[Action]:
        return [self.Noop()]

    def Noop(self) -> Action:
        return Action(name="noop")


class MoveActionConfig(ActionConfig):
    action_handler: str = Field(default="move")
    allowed_directions: list[Direction] = Field(default_factory=lambda: list(CardinalDirections))

    def _actions(self) -> list[Action]:
        return [self.Move(d) for d in self.allowed_directions]

    def Move(self, direction: str) -> Action:
        return Action(name=f"move_{direction}")


class ChangeVibeActionConfig(ActionConfig):
    model_config = ConfigDict(extra="forbid", arbitrary_types_allowed=True)
    action_handler: str = Field(default="change_vibe")
    vibes: list[Vibe] = Field(default_factory=lambda: list(VIBES))

    @model_validator(mode="before")
    @classmethod
    def _coerce_vibes(cls, data: Any) -> Any:
        if isinstance(data, dict) and "vibes" in data:
            out = []
            for v in data["vibes"]:
                if isinstance(v, Vibe):
                    out.append(v)
                elif isinstance(v, str):
                    out.append(VIBE_BY_NAME.get(v, Vibe("", v)))
                elif isinstance(v, dict):
                    out.append(Vibe(v.get("symbol", ""), v["name"], v.get("category", "misc")))
                else:
                    raise TypeError(f"cannot interpret {v!r} as a Vibe")
            data = dict(data)
            data["vibes"] = out
        return data

    def _actions(self) -> list[Action]:
        return [self.ChangeVibe(v) for v in self.vibes]

    def ChangeVibe(self, vibe: Vibe) -> Action:
        return Action(name=f"change_vibe_{vibe.name}")


class AttackOutcome(Config):
    actor_inv_delta: dict[str, int] = Field(default_factory=dict)
    target_inv_delta: dict[str, int] = Field(default_factory=dict)
    loot: list[str] = Field(default_factory=list)
    freeze: int = Field(default=0)


class AttackActionConfig(ActionConfig):
    action_handler: str = Field(default="attack")
    defense_resources: dict[str, int] = Field(default_factory=dict)
    armor_resources: dict[str, int] = Field(default_factory=dict)
    weapon_resources: dict[str, int] = Field(default_factory=dict)
    success: AttackOutcome = Field(default_factory=AttackOutcome)
    vibes: list[str] = Field(default_factory=list)
    vibe_bonus: dict[str, int] = Field(default_factory=dict)

    def _actions(self) -> list[Action]:
        return []  # attack only fires from move (reference attack.hpp:87-90)


class VibeTransfer(Config):
    vibe: str
    target: dict[str, int] = Field(default_factory=dict)
    actor: dict[str, int] = Field(default_factory=dict)


class TransferActionConfig(ActionConfig):
    action_handler: str = Field(default="transfer")
    vibe_transfers: list[VibeTransfer] = Field(default_factory=list)

    def _actions(self) -> list[Action]:
        return []  # transfer only fires from move (reference transfer.hpp:56-59)


class ActionsConfig(Config):
    noop: NoopActionConfig = Field(default_factory=lambda: NoopActionConfig())
    move: MoveActionConfig = Field(default_factory=lambda: MoveActionConfig())
    attack: AttackActionConfig = Field(default_factory=lambda: AttackActionConfig(enabled=False))
    transfer: TransferActionConfig = Field(default_factory=lambda: TransferActionConfig(enabled=False))
    change_vibe: ChangeVibeActionConfig = Field(default_factory=lambda: ChangeVibeActionConfig())

    def actions(self) -> list[Action]:
        # Fixed order: noop, move dirs, [attack: none], [transfer: none], change_vibe_*
        # (reference: mettagrid_config.py:323-327 and BIND/mettagrid_c.cpp:291-352).
        out: list[Action] = []
        for a in (self.noop, self.move, self.attack, self.transfer, self.change_vibe):
            out.extend(a.actions())
        return out


class GlobalObsConfig(Config):
    episode_completion_pct: bool = Field(default=True)
    last_action: bool = Field(default=True)
    last_reward: bool = Field(default=True)
    compass: bool = Field(default=False)
    goal_obs: bool = Field(default=False)


class GridObjectConfig(Config):
    name: str
    map_name: str = Field(default="")
    render_name: str = Field(default="")
    render_symbol: str = Field(default="?")
    tags: list[str] = Field(default_factory=list)
    vibe: int = Field(default=0, ge=0, le=255)
    collective: Optional[str] = Field(default=None)

    @model_validator(mode="after")
    def _defaults_from_name(self) -> "GridObjectConfig":
        # object.__setattr__ avoids re-entering validation (validate_assignment is on).
        if not self.map_name:
            object.__setattr__(self, "map_name", self.name)
        if not self.render_name:
            object.__setattr__(self, "render_name", self.name)
        if not self.tags:
            object.__setattr__(self, "tags", [self.render_name])
        if self.collective:
            tag = f"collective:{self.collective}"
            if tag not in self.tags:
                object.__setattr__(self, "tags", self.tags + [tag])
        return self


class WallConfig(GridObjectConfig):
    pydantic_type: Literal["wall"] = "wall"
    name: str = Field(default="wall")


class ProtocolConfig(Config):
    min_agents: int = Field(default=0, ge=0)
    vibes: list[str] = Field(default_factory=list)
    input_resources: dict[str, int] = Field(default_factory=dict)
    output_resources: dict[str, int] = Field(default_factory=dict)
    cooldown: int = Field(ge=0, default=0)


class AssemblerConfig(GridObjectConfig):
    pydantic_type: Literal["assembler"] = "assembler"
    protocols: list[ProtocolConfig] = Field(default_factory=list)
    allow_partial_usage: bool = Field(default=False)
    max_uses: int = Field(default=0, ge=0)
    clip_immune: bool = Field(default=False)
    start_clipped: bool = Field(default=False)
    chest_search_distance: int = Field(default=0, ge=0)


class ChestConfig(GridObjectConfig):
    pydantic_type: Literal["chest"] = "chest"
    name: str = Field(default="chest")
    vibe_transfers: dict[str, dict[str, int]] = Field(default_factory=dict)
    inventory: InventoryConfig = Field(default_factory=InventoryConfig)


class ClipperConfig(Config):
    unclipping_protocols: list[ProtocolConfig] = Field(default_factory=list)
    length_scale: int = Field(default=0, ge=0)
    scaled_cutoff_distance: int = Field(default=3, ge=1)
    clip_period: int = Field(default=0, ge=0)


class CollectiveConfig(Config):
    name: str
    inventory: InventoryConfig = Field(default_factory=InventoryConfig)


AnyGridObjectConfig = Annotated[
    Union[WallConfig, AssemblerConfig, ChestConfig],
    Field(discriminator="pydantic_type"),
]

DEFAULT_RESOURCE_NAMES = [
    "ore_red",
    "ore_blue",
    "ore_green",
    "battery_red",
    "battery_blue",
    "battery_green",
    "heart",
    "armor",
    "laser",
    "blueprint",
]


class GameConfig(Config):
    model_config = ConfigDict(extra="forbid", arbitrary_types_allowed=True, validate_assignment=False)

    resource_names: list[str] = Field(default_factory=lambda: list(DEFAULT_RESOURCE_NAMES))
    vibe_names: list[str] = Field(default_factory=list)
    num_agents: int = Field(ge=1, default=24)
    max_steps: int = Field(ge=0, default=10000)
    episode_truncates: bool = Field(default=False)
    obs: ObsConfig = Field(default_factory=ObsConfig)
    agent: AgentConfig = Field(default_factory=AgentConfig)
    agents: list[AgentConfig] = Field(default_factory=list)
    actions: ActionsConfig = Field(default_factory=lambda: ActionsConfig())
    global_obs: GlobalObsConfig = Field(default_factory=GlobalObsConfig)
    objects: dict[str, AnyGridObjectConfig] = Field(default_factory=dict)
    params: Optional[Any] = None
    inventory_regen_interval: int = Field(default=0, ge=0)
    clipper: Optional[ClipperConfig] = Field(default=None)
    collectives: list[CollectiveConfig] = Field(default_factory=list)
    # The reference types this as AnyMapBuilderConfig; here any object with `.build(num_agents, rng)`.
    map_builder: Any = Field(default=None)
    protocol_details_obs: bool = Field(default=True)
    reward_estimates: Optional[dict[str, float]] = Field(default=None)

    @model_validator(mode="after")
    def _sync_vibe_names(self) -> "GameConfig":
        self.vibe_names = [v.name for v in self.actions.change_vibe.vibes]
        return self

    def id_map(self):
        from metta_amd.id_map import IdMap

        return IdMap(self)


class MettaGridConfig(Config):
    model_config = ConfigDict(extra="forbid", arbitrary_types_allowed=True, validate_assignment=False)

    label: str = Field(default="mettagrid")
    game: GameConfig = Field(default_factory=GameConfig)
    desync_episodes: bool = Field(default=True)

    def with_ascii_map(self, map_data: list[list[str]], char_to_map_name: dict[str, str]) -> "MettaGridConfig":
        from metta_amd.mapgen import AsciiMapBuilder

        self.game.map_builder = AsciiMapBuilder(map_data=map_data, char_to_map_name=char_to_map_name)
        return self

    @staticmethod
    def EmptyRoom(
        num_agents: int, width: int = 10, height: int = 10, border_width: int = 1, with_walls: bool = False
    ) -> "MettaGridConfig":
        from metta_amd.mapgen import RandomMapBuilder

        objects: dict[str, Any] = {}
        if border_width > 0 or with_walls:
            objects["wall"] = WallConfig()
        return MettaGridConfig(
            game=GameConfig(
                map_builder=RandomMapBuilder(agents=num_agents, width=width, height=height, border_width=border_width),
                actions=ActionsConfig(move=MoveActionConfig(), change_vibe=ChangeVibeActionConfig()),
                num_agents=num_agents,
                objects=objects,
            )
        )
