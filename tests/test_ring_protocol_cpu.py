"""A model of das_f64_ring_kernel's slot protocol (beamform_amd/csrc/das_f64_w64.hip, LAYOUT 1): eight wavefronts draw frame pairs in order,
move each pair's two new hops into slots 2u, 2u + 1 (mod R) of the block's ring and read the hop in front of the pair out of the previous
pair's second slot.  A slot's state gains 4 per generation (published 2, readers 1 + 1 -- or 2 where there is one reader); a writer waits
for 4 g, the reader of a neighbour's slot for 4 g + 2; a chunk's first pair uses a private slot.  The model runs the protocol under random
interleavings and checks what the kernel relies on: no deadlock (a waiting wavefront holds nothing the wait is for: it has released its
previous pair), no slot rewritten before every reader of its previous content is done, no slot read before it is published, and every
state word ends at 4 x generations."""
import random

import pytest

R = 32  # kRingR


def run(chunks, n_waves, seed, slow_wave=None):
    rng = random.Random(seed)
    pairs = [(pos, ln) for ln in chunks for pos in range(ln)]  # virtual index u -> (position in its chunk, chunk length)
    state = [0] * R
    content = [None] * R       # which hop-number (2u or 2u + 1) a slot holds, None while being written
    readers_left = [0] * R     # readers of the current content that have not released yet
    nxt = 0
    # per wavefront: (phase, u); phases: draw -> wait_free -> publish -> wait_prev -> compute -> release
    waves = [["draw", None] for _ in range(n_waves)]
    done, idle_ticks = 0, 0
    while done < len(pairs):
        w = rng.randrange(n_waves)
        ph, u = waves[w]
        progressed = True
        if ph == "draw":
            if nxt >= len(pairs):
                progressed = False
            else:
                waves[w] = ["wait_free", nxt]
                nxt += 1
        elif ph == "wait_free":
            sA, g4 = (2 * u) % R, 4 * ((2 * u) // R)
            if state[sA] >= g4 and state[sA + 1] >= g4:
                assert state[sA] == g4 and state[sA + 1] == g4          # exactly the previous generation's total
                assert readers_left[sA] == 0 and readers_left[sA + 1] == 0   # nobody still reads what is about to be overwritten
                content[sA] = content[sA + 1] = None
                waves[w] = ["publish", u]
            else:
                progressed = False
        elif ph == "publish":
            pos, ln = pairs[u]
            sA = (2 * u) % R
            content[sA], content[sA + 1] = 2 * u, 2 * u + 1
            readers_left[sA] = 1
            readers_left[sA + 1] = 2 if pos + 1 < ln else 1
            state[sA] += 2
            state[sA + 1] += 2
            waves[w] = ["wait_prev", u]
        elif ph == "wait_prev":
            pos, _ = pairs[u]
            if pos == 0:
                waves[w] = ["compute", u]
            else:
                jp = 2 * u - 1
                if state[jp % R] - 4 * (jp // R) >= 2:
                    assert content[jp % R] == jp                          # published, and still the hop this pair needs
                    waves[w] = ["compute", u]
                else:
                    progressed = False
        elif ph == "compute":
            if rng.random() < (0.002 if w == slow_wave else 0.3):   # (a pair takes a random number of ticks; one wavefront may crawl)
                waves[w] = ["release", u]
        else:  # release
            pos, ln = pairs[u]
            sA = (2 * u) % R
            state[sA] += 2
            readers_left[sA] -= 1
            state[sA + 1] += 1 if pos + 1 < ln else 2
            readers_left[sA + 1] -= 1
            if pos > 0:
                jp = 2 * u - 1
                assert content[jp % R] == jp
                state[jp % R] += 1
                readers_left[jp % R] -= 1
            waves[w] = ["draw", None]
            done += 1
        idle_ticks = 0 if progressed else idle_ticks + 1
        assert idle_ticks < 100000, ("deadlock", waves, nxt)
    n_hops = 2 * len(pairs)
    for s in range(R):
        gens = (n_hops - s + R - 1) // R if n_hops > s else 0
        assert state[s] == 4 * gens, (s, state[s], gens)


@pytest.mark.parametrize("seed", range(8))
def test_ring_slot_protocol_under_random_interleavings(seed):
    rng = random.Random(100 + seed)
    chunks = [104] + [rng.choice([1, 2, 3, 5, 8, 13]) for _ in range(40)]   # a long first chunk, then the small ones of the guided plan
    run(chunks, 8, seed)
    run([1] * 70, 8, seed)        # every pair opens a chunk: private slots only, no neighbour reads
    run([200], 3, seed)           # fewer wavefronts than the ring has pairs
    run([150] + [4] * 10, 8, seed, slow_wave=seed % 8)   # one wavefront crawls: the others run into the ring's 16-pair window and wait for it
