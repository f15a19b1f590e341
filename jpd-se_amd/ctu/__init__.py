"""MI355X-native drop-in for the `ctu` package surface of SenseBrain/JPD-SE that lies on the
training hot path: `ctu.models`, `ctu.trainers` and the two `ctu.utils` helpers they use
(SURVEY.md §8b).  Everything else of the reference (parsers, datasets, visualisation,
quantizers) is out of scope and intentionally absent."""
