// ref_state_shim.cpp -- C-ABI wrapper around the REFERENCE's own State class.
//
// TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists, by
// oracle/Makefile target `ref`, together with the reference's unmodified
// mcts/src/state.cpp compiled from where it lies (no reference source is
// copied into this repository).  Output goes to oracle/_ref/ (git-ignored).
// It is used to generate tests/golden/rules_*.json and to check the oracle's
// rules restatement move for move.
//
// Board size is a compile-time macro in the reference (mcts/CMakeLists.txt:36),
// so one library is built per shape: -DSHAPE=<S> -DSIZE=<S*S>.
#include "state.h" // /root/reference/mcts/src/state.h via -I

extern "C" {

int ref_shape(void) { return SHAPE; }

// Plays `moves` from the empty board with State::make_move (state.cpp:81-110).
// Per ply records player, move_cnt, winner.  Returns the index of the first
// move that threw ("Invalid move ...", state.cpp:85-86), else n.
int ref_play(const int *moves, int n, int *player, int *move_cnt, float *winner)
{
    State s;
    int i;
    for (i = 0; i < n; i++) {
        try {
            s.make_move(moves[i]);
        } catch (const std::runtime_error &) {
            break;
        }
        player[i] = s.player;
        move_cnt[i] = s.move_cnt;
        winner[i] = s.winner;
    }
    return i;
}

// Same, but the caller may force the side to move before each ply
// (Cmcts::set_player, cmcts.cpp:48): force[i] < 0 leaves it alone.
int ref_play_forced(const int *moves, const int *force, int n, int *player, int *move_cnt, float *winner)
{
    State s;
    int i;
    for (i = 0; i < n; i++) {
        if (force[i] >= 0)
            s.player = force[i];
        try {
            s.make_move(moves[i]);
        } catch (const std::runtime_error &) {
            break;
        }
        player[i] = s.player;
        move_cnt[i] = s.move_cnt;
        winner[i] = s.winner;
    }
    return i;
}

// State::get_board (state.cpp:39-48) after playing `moves`; needs the calling
// Python interpreter's GIL (load with ctypes.PyDLL).  out: 2*SIZE floats.
int ref_get_board(const int *moves, int n, float *out)
{
    State s;
    for (int i = 0; i < n; i++)
        s.make_move(moves[i]);
    py::array_t<float> b = s.get_board();
    auto r = b.unchecked<3>();
    for (int p = 0; p < 2; p++)
        for (int y = 0; y < SHAPE; y++)
            for (int x = 0; x < SHAPE; x++)
                out[p * SIZE + y * SHAPE + x] = r(p, y, x);
    return 0;
}

// State::is_valid (state.cpp:29-37) after playing `moves`.
int ref_is_valid(const int *moves, int n, int action)
{
    State s;
    for (int i = 0; i < n; i++)
        s.make_move(moves[i]);
    return s.is_valid(action) ? 1 : 0;
}

} // extern "C"
