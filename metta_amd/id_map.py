"""Observation feature ids (reference: PY/config/id_map.py:91-180).

Ids are positional: 12 fixed features, then per resource `inv:<r>` followed by its power tokens
`inv:<r>:p1..`, then (if protocol_details_obs) `protocol_input:<r>` for all r, `protocol_output:<r>` for all r.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

FIXED_FEATURES: list[tuple[str, float]] = [
    ("agent:group", 10.0),
    ("agent:frozen", 1.0),
    ("episode_completion_pct", 255.0),
    ("last_action", 10.0),
    ("last_reward", 100.0),
    ("goal", 100.0),
    ("vibe", 255.0),
    ("agent:compass", 1.0),
    ("tag", 10.0),
    ("cooldown_remaining", 255.0),
    ("clipped", 1.0),
    ("remaining_uses", 255.0),
]


def num_inventory_tokens_needed(max_inventory_value: int, token_value_base: int) -> int:
    """Feature-id side: ceil(log_base(max+1)) in floating point, exactly as id_map.py:22-35 (the ids are
    assigned by the Python layer, so its float rounding is part of the contract)."""
    if max_inventory_value == 0:
        return 1
    return math.ceil(math.log(max_inventory_value + 1, token_value_base))


def encoder_num_inventory_tokens(max_inventory_value: int, token_value_base: int) -> int:
    """Encoder side: integer digit count, as ObservationEncoder::compute_num_tokens
    (observation_encoder.hpp:90-105)."""
    if max_inventory_value == 0:
        return 1
    n, v = 0, max_inventory_value
    while v > 0:
        v //= token_value_base
        n += 1
    return n


@dataclass(frozen=True)
class ObservationFeatureSpec:
    id: int
    name: str
    normalization: float


class IdMap:
    def __init__(self, config):
        self._config = config
        self._features: list[ObservationFeatureSpec] | None = None

    def features(self) -> list[ObservationFeatureSpec]:
        if self._features is None:
            self._features = self._compute_features()
        return self._features

    def feature_id(self, name: str) -> int:
        for f in self.features():
            if f.name == name:
                return f.id
        raise KeyError(f"Unknown observation feature: {name}")

    def feature(self, name: str) -> ObservationFeatureSpec:
        for f in self.features():
            if f.name == name:
                return f
        raise KeyError(f"Unknown observation feature: {name}")

    def tag_names(self) -> list[str]:
        cfg = self._config
        return sorted(
            set(
                [t for o in cfg.objects.values() for t in o.tags]
                + [t for a in cfg.agents for t in a.tags]
                + list(cfg.agent.tags)
            )
        )

    def _compute_features(self) -> list[ObservationFeatureSpec]:
        cfg = self._config
        feats: list[ObservationFeatureSpec] = []
        for name, norm in FIXED_FEATURES:
            feats.append(ObservationFeatureSpec(len(feats), name, norm))
        base = cfg.obs.token_value_base
        n_inv = num_inventory_tokens_needed(65535, base)
        for r in cfg.resource_names:
            feats.append(ObservationFeatureSpec(len(feats), f"inv:{r}", float(base)))
            for p in range(1, n_inv):
                feats.append(ObservationFeatureSpec(len(feats), f"inv:{r}:p{p}", float(base)))
        if cfg.protocol_details_obs:
            for r in cfg.resource_names:
                feats.append(ObservationFeatureSpec(len(feats), f"protocol_input:{r}", 100.0))
            for r in cfg.resource_names:
                feats.append(ObservationFeatureSpec(len(feats), f"protocol_output:{r}", 100.0))
        return feats
