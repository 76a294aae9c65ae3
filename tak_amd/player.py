"""`Player` (reference alpha-tak/src/player.rs:22-199) on top of the engine's search entry points — the API shape bots and
the analysis tools use for ONE game: rollout / add_noise / pick_move / play_move / get_examples.

The reference overlaps a rollout thread with the network call (one batch of virtual rollouts always in flight).  Here the
tree lives on the GPU and `rollout` is one iteration of the one-game search with TgSearchConfig.batch = `batch`: `batch` virtual
rollouts in the tree, one network call for the leaves that are not terminal, de-virtualisation in order — the same tree
updates as Player::rollout (player.rs:77-110, 125-128), without the thread.  The engine's max_batch must be ≥ batch.  Several
Players can share one engine only one at a time (the engine holds one search state)."""
import numpy as np

from .engine import TG_MAX_MOVES


class Player:
    def __init__(self, engine, batch, save_examples, game, arena_nodes=1 << 16, seed=0):
        """Player::new(network, batch, save_examples, create_analysis = false, &game); `game` is a packed state."""
        self.e = engine
        self.batch = int(batch)
        self.save_examples = bool(save_examples)
        self.examples = []  # IncompleteExample: (state, moves, visits)
        self.rng = np.random.default_rng(seed)
        engine.search_create(1, arena_nodes=arena_nodes, seed=seed, batch=self.batch)
        engine.search_reset(np.ascontiguousarray(game, np.uint8).reshape(1, -1))
        self.rollout()  # the reference requests the first batch in the constructor (player.rs:65-66)

    def state(self):
        return self.e.search_states()[0]

    def rollout(self, game=None):
        """One batch of `batch` virtual rollouts from the current root (player.rs:125-128)."""
        self.e.search_run(1)

    def add_noise(self, alpha, ratio, game=None):
        """Node::apply_dirichlet on the root (player.rs:118-122)."""
        self.e.search_apply_dirichlet(alpha, ratio)

    def improved_policy(self):
        """Node::improved_policy: [(move, visits)] of the root's children."""
        r = self.e.search_root()
        c = int(r["counts"][0])
        return r["moves"][0, :c].copy(), r["visits"][0, :c].copy()

    def pick_move(self, exploitation):
        """Node::pick_move (play.rs:49-67): most visited (last on ties) or sampled ∝ visits."""
        moves, visits = self.improved_policy()
        if len(moves) == 0:
            raise RuntimeError("pick_move on a root without children")
        if exploitation:
            return int(moves[len(visits) - 1 - int(np.argmax(visits[::-1]))])
        total = int(visits.sum())
        if total == 0:
            raise RuntimeError("pick_move: no visits to sample from")  # WeightedIndex panics in the reference
        return int(moves[int(self.rng.choice(len(moves), p=visits / total))])

    def play_move(self, move, game=None, with_info=True):
        """Advance the tree (tree reuse) and the game; record an IncompleteExample (player.rs:136-166)."""
        if self.save_examples and with_info:
            moves, visits = self.improved_policy()
            self.examples.append((self.state().copy(), moves, visits))
        self.e.search_play(np.array([move], np.uint16))
        self.rollout()  # refill: the new root is expanded like the batch the reference keeps in flight

    def get_examples(self, result):
        """Complete the collected examples with the game result (TgResult code) from each mover's perspective
        (player.rs:170-193) → (states, n_moves, moves, visits, results) in the layout of tg_selfplay_drain / tg_train."""
        if result == 0:
            raise ValueError("cannot complete examples with an ongoing game")
        white = 1.0 if result in (1, 2) else -1.0 if result in (3, 4) else 0.0
        k = len(self.examples)
        sb = self.e.sb
        states = np.zeros((k, sb), np.uint8)
        n_moves = np.zeros(k, np.int32)
        moves = np.zeros((k, TG_MAX_MOVES), np.uint16)
        visits = np.zeros((k, TG_MAX_MOVES), np.uint32)
        results = np.zeros(k, np.float32)
        for i, (st, mv, vs) in enumerate(self.examples):
            states[i] = st
            n_moves[i] = len(mv)
            moves[i, : len(mv)] = mv
            visits[i, : len(mv)] = vs
            to_move = st[sb - 16 + 1]
            results[i] = white if to_move == 0 else -white
        self.examples = []
        return states, n_moves, moves, visits, results
