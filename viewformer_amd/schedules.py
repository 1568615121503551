"""String-parsable scalar schedules of the transformer's ``localization_weight`` (``config.json`` stores them as text).

Mirror of the grammar and semantics of viewformer/utils/schedules.py:72-248 (``Schedule.from_str`` :97-103, ``with_total_steps``
:114-118, ``is_zero`` :120-121,148-149,183-184,212-213,236-237): a constant (``'1'``, ``'5.'``), ``linear(a,b[,N])``,
``cosine(a,b[,N])`` — N omitted is completed with the model's ``total_steps`` (migt.py:268) — and ``warmup(<inner>,W)``.
``MIGT.use_localization`` is ``not schedule.is_zero()`` (migt.py:269).  Host-side scalar math only.
"""
import math
from dataclasses import dataclass, replace
from typing import Optional


class Schedule:
    def __call__(self, t) -> float:
        raise NotImplementedError

    def is_zero(self) -> bool:
        return False

    def with_total_steps(self, n):
        return self


@dataclass(frozen=True)
class Constant(Schedule):
    value: float

    def __call__(self, t):
        return float(self.value)

    def is_zero(self):
        return self.value == 0

    def __str__(self):
        return str(self.value)


@dataclass(frozen=True)
class _Ramp(Schedule):
    initial_value: float
    final_value: float
    num_total_steps: Optional[int] = None
    kind = ''

    def is_zero(self):
        return self.initial_value == self.final_value == 0

    def with_total_steps(self, n):
        return self if self.num_total_steps is not None else replace(self, num_total_steps=n)

    def _frac(self, t):
        if self.num_total_steps is None:
            raise ValueError(f'{self}: no step count (give N or call with_total_steps)')
        return min(1.0, float(t) / self.num_total_steps)

    def __str__(self):
        return f'{self.kind}({self.initial_value},{self.final_value},{self.num_total_steps})'


class Linear(_Ramp):
    kind = 'linear'

    def __call__(self, t):                                   # schedules.py:170-171
        return self.initial_value + self._frac(t) * (self.final_value - self.initial_value)


class Cosine(_Ramp):
    kind = 'cosine'

    def __call__(self, t):                                   # schedules.py:201-203
        return self.final_value + (self.initial_value - self.final_value) * 0.5 * (math.cos(self._frac(t) * math.pi) + 1)


@dataclass(frozen=True)
class Warmup(Schedule):
    inner: Schedule
    warmup_steps: int

    def __call__(self, t):                                   # schedules.py:222-225
        w = min(float(t), self.warmup_steps)
        rest = max(float(t) - self.warmup_steps, 0)
        return (w / self.warmup_steps) * self.inner(rest)

    def is_zero(self):
        return self.inner.is_zero()

    def with_total_steps(self, n):
        return replace(self, inner=self.inner.with_total_steps(n))

    def __str__(self):
        return f'warmup({self.inner},{self.warmup_steps})'


def parse(value) -> Schedule:
    """text (or number, or Schedule) -> Schedule; raises ValueError on anything the reference's grammar does not produce"""
    if isinstance(value, Schedule):
        return value
    s = str(value).strip()
    if s.startswith('warmup(') and s.endswith(')') and ',' in s:
        body = s[len('warmup('):-1]
        cut = body.rindex(',')
        return Warmup(parse(body[:cut]), int(body[cut + 1:].strip()))
    for cls in (Cosine, Linear):
        if s.startswith(cls.kind + '(') and s.endswith(')'):
            parts = [p.strip() for p in s[len(cls.kind) + 1:-1].split(',')]
            if len(parts) not in (2, 3):
                raise ValueError(f'{cls.kind} schedule takes 2 or 3 arguments: {s!r}')
            n = None if len(parts) == 2 or parts[2] in ('None', '') else int(float(parts[2]))
            return cls(float(parts[0]), float(parts[1]), n)
    try:
        return Constant(float(s))
    except ValueError:
        raise ValueError(f'cannot parse schedule {s!r}') from None
