/*
 * az_oracle.c -- CPU restatement of the reference self-play hot path.
 * TEST INFRASTRUCTURE ONLY (see az_oracle.h).  Each function cites the
 * reference lines it follows.  Deliberately written the way the reference is
 * (history lists, 69 line masks, object-per-node tree) and NOT the way the HIP
 * engine is (bitboards, pooled nodes), so that agreement between the two is
 * evidence and not tautology.
 */
#include "az_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== */
/* Connect Four constants: connectfour.py:11-21                           */
/* ===================================================================== */
#define C4_W 7
#define C4_H 6
#define C4_STREAK 4
#define C4_SPACES (C4_W * C4_H)
#define C4_NUM_WINS 69

static uint8_t c4_wins[C4_NUM_WINS][C4_SPACES];
static int c4_wins_ready = 0;

/* connectfour.py:23-44 -- same enumeration order as the reference loop */
static void c4_build_wins(void) {
    if (c4_wins_ready) return;
    memset(c4_wins, 0, sizeof(c4_wins));
    int n = 0;
    for (int row = 0; row < C4_H; ++row) {
        for (int col = 0; col < C4_W; ++col) {
            int l_right = col <= C4_W - C4_STREAK;
            int l_above = row <= C4_H - C4_STREAK;
            if (l_right) {
                for (int i = 0; i < C4_STREAK; ++i) c4_wins[n][row * C4_W + col + i] = 1;
                n++;
            }
            if (l_above) {
                for (int i = 0; i < C4_STREAK; ++i) c4_wins[n][(row + i) * C4_W + col] = 1;
                n++;
            }
            if (l_right && l_above) {
                for (int i = 0; i < C4_STREAK; ++i) {
                    c4_wins[n + 0][(row + i) * C4_W + col + i] = 1;
                    c4_wins[n + 1][(row + C4_STREAK - i - 1) * C4_W + col + i] = 1;
                }
                n += 2;
            }
        }
    }
    c4_wins_ready = 1;
}

/* tictactoe.py:14-23 */
static const int ttt_wins[8][3] = {
    {0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6}, {1, 4, 7}, {2, 5, 8}, {0, 4, 8}, {2, 4, 6}};

int orc_num_actions(int game) { return game == ORC_C4 ? C4_W : 9; }
int orc_max_moves(int game) { return game == ORC_C4 ? C4_SPACES : 9; }
void orc_board_shape(int game, int* h, int* w) {
    if (game == ORC_C4) { *h = C4_H; *w = C4_W; } else { *h = 3; *w = 3; }
}

int orc_winning_positions(int game, uint8_t* out) {
    if (game == ORC_C4) {
        c4_build_wins();
        memcpy(out, c4_wins, sizeof(c4_wins));
        return C4_NUM_WINS;
    }
    memset(out, 0, 8 * 9);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 3; ++j) out[i * 9 + ttt_wins[i][j]] = 1;
    return 8;
}

/* game.py:23-28 */
static int next_player(int len) { return len % 2; }
static int prev_player(int len) { return 1 - next_player(len); }

/* connectfour.py:80-81 ; tictactoe.py:47-48 */
int orc_valid_actions(int game, const uint8_t* hist, int len, int* out) {
    int n = 0;
    if (game == ORC_C4) {
        for (int i = 0; i < C4_W; ++i) {
            int count = 0;
            for (int k = 0; k < len; ++k) count += (hist[k] == i);
            if (count < C4_H) out[n++] = i;
        }
    } else {
        for (int i = 0; i < 9; ++i) {
            int played = 0;
            for (int k = 0; k < len; ++k) played |= (hist[k] == i);
            if (!played) out[n++] = i;
        }
    }
    return n;
}

/* connectfour.py:112-120 ; tictactoe.py:50-62 */
void orc_render(int game, const uint8_t* hist, int len, float* out) {
    if (game == ORC_C4) {
        memset(out, 0, sizeof(float) * 2 * C4_SPACES);
        int num_filled[C4_W] = {0};
        int np_ = next_player(len);
        for (int i = 0; i < len; ++i) {
            int col = hist[i];
            int action_player = i % 2;
            int j = (action_player == np_) ? 0 : 1;
            if (num_filled[col] < C4_H) out[j * C4_SPACES + num_filled[col] * C4_W + col] = 1.0f;
            num_filled[col] += 1;
        }
    } else {
        memset(out, 0, sizeof(float) * 2 * 9);
        int np_ = next_player(len), pp = prev_player(len);
        for (int i = np_; i < len; i += 2) out[0 * 9 + hist[i]] = 1.0f;
        for (int i = pp; i < len; i += 2) out[1 * 9 + hist[i]] = 1.0f;
    }
}

/* connectfour.py:92-110 ; tictactoe.py:36-44 */
int orc_winner(int game, const uint8_t* hist, int len) {
    if (game == ORC_C4) {
        if (len < 2 * C4_STREAK - 1) return ORC_WINNER_NONE;
        c4_build_wins();
        float board[2 * C4_SPACES];
        orc_render(game, hist, len, board);
        const float* prev = board + C4_SPACES; /* render()[1] */
        float best = 0.0f;
        for (int w = 0; w < C4_NUM_WINS; ++w) {
            float s = 0.0f;
            for (int c = 0; c < C4_SPACES; ++c) s += (float)c4_wins[w][c] * prev[c];
            if (s > best) best = s;
        }
        if (best >= (float)C4_STREAK) return prev_player(len);
        if (len == C4_SPACES) return ORC_WINNER_DRAW;
        return ORC_WINNER_NONE;
    }
    if (len < 5) return ORC_WINNER_NONE;
    for (int player = 0; player < 2; ++player) {
        for (int w = 0; w < 8; ++w) {
            int hit[3] = {0, 0, 0};
            for (int i = player; i < len; i += 2)
                for (int j = 0; j < 3; ++j)
                    if (hist[i] == ttt_wins[w][j]) hit[j] = 1;
            if (hit[0] + hit[1] + hit[2] == 3) return player;
        }
    }
    if (len == 9) return ORC_WINNER_DRAW;
    return ORC_WINNER_NONE;
}

/* connectfour.py:84-89 ; tictactoe.py:26-33 */
int orc_terminal(int game, const uint8_t* hist, int len) {
    if (game == ORC_C4) {
        if (len < 2 * C4_STREAK - 1) return 0;
        if (len == C4_SPACES) return 1;
        return orc_winner(game, hist, len) != ORC_WINNER_NONE;
    }
    if (len < 5) return 0;
    if (len == 9) return 1;
    return orc_winner(game, hist, len) != ORC_WINNER_NONE;
}

/* game.py:42-47 (caller guarantees a winner exists) */
int orc_terminal_value(int game, const uint8_t* hist, int len, int player) {
    int w = orc_winner(game, hist, len);
    if (w < 0) return 0;
    return player == w ? +1 : -1;
}

/* ===================================================================== */
/* MCTreeNode: mcts.py:5-91                                               */
/* ===================================================================== */
struct orc_node {
    orc_node*  parent;
    orc_node** children; /* NULL until expanded */
    int        num;      /* number of edges */
    uint32_t   N[ORC_MAX_ACTIONS];
    float      W[ORC_MAX_ACTIONS];
    float      P[ORC_MAX_ACTIONS];
};

orc_node* orc_node_new(orc_node* parent) { /* mcts.py:11-14 */
    orc_node* n = (orc_node*)calloc(1, sizeof(orc_node));
    n->parent = parent;
    return n;
}

void orc_node_free(orc_node* n) {
    if (!n) return;
    if (n->children) {
        for (int i = 0; i < n->num; ++i) orc_node_free(n->children[i]);
        free(n->children);
    }
    free(n);
}

int orc_node_expanded(const orc_node* n) { return n->children != NULL; } /* mcts.py:16-18 */
int orc_node_num_edges(const orc_node* n) { return n->children ? n->num : 0; }

void orc_node_expand(orc_node* n, const float* p, int num) { /* mcts.py:20-26 */
    if (n->children) {
        for (int i = 0; i < n->num; ++i) orc_node_free(n->children[i]);
        free(n->children);
    }
    n->num = num;
    n->children = (orc_node**)calloc((size_t)(num > 0 ? num : 1), sizeof(orc_node*));
    for (int i = 0; i < num; ++i) { n->N[i] = 0; n->W[i] = 0.0f; n->P[i] = p[i]; }
}

orc_node* orc_node_traverse(orc_node* n, int ai) { /* mcts.py:28-32 */
    if (n->children[ai] == NULL) n->children[ai] = orc_node_new(n);
    return n->children[ai];
}

orc_node* orc_node_child(const orc_node* n, int ai) { return n->children ? n->children[ai] : NULL; }
orc_node* orc_node_parent(const orc_node* n) { return n->parent; }

void orc_node_kill_siblings(orc_node* n) { /* mcts.py:76-82 */
    if (n->parent == NULL) return;
    orc_node* p = n->parent;
    for (int i = 0; i < p->num; ++i) {
        if (p->children[i] != n) { orc_node_free(p->children[i]); p->children[i] = NULL; }
    }
}

void orc_node_edges(const orc_node* n, uint32_t* N, float* W, float* P) {
    for (int i = 0; i < n->num; ++i) { if (N) N[i] = n->N[i]; if (W) W[i] = n->W[i]; if (P) P[i] = n->P[i]; }
}
void orc_node_set_edges(orc_node* n, const uint32_t* N, const float* W, const float* P) {
    for (int i = 0; i < n->num; ++i) { if (N) n->N[i] = N[i]; if (W) n->W[i] = W[i]; if (P) n->P[i] = P[i]; }
}

/* mcts.py:43-50: float32 array divided in place by a uint32 array -> the
 * quotient is formed in double and rounded to float */
void orc_node_mean_value(const orc_node* n, float* Q) {
    for (int i = 0; i < n->num; ++i)
        Q[i] = n->N[i] != 0 ? (float)((double)n->W[i] / (double)n->N[i]) : 0.0f;
}

/* mcts.py:52-59 */
void orc_node_ucb_score(const orc_node* n, double c_base, double c_init,
                        const orc_puct_entry* lut, int lut_len, double* out) {
    float Q[ORC_MAX_ACTIONS];
    orc_node_mean_value(n, Q);
    uint64_t Ns = 0;
    for (int i = 0; i < n->num; ++i) Ns += n->N[i];
    double Cs, sq;
    if (lut && Ns < (uint64_t)lut_len) { Cs = lut[Ns].cs; sq = lut[Ns].sqrt_ns; }
    else { Cs = log((double)(1 + Ns + (uint64_t)c_base) / c_base) + c_init; sq = sqrt((double)Ns); }
    for (int i = 0; i < n->num; ++i) {
        double s = sq / (double)(uint32_t)(1u + n->N[i]);
        double U = (Cs * (double)n->P[i]) * s;
        out[i] = (double)Q[i] + U;
    }
}

static int argmax_first_d(const double* x, int n) { /* np.argmax: first maximal index */
    int b = 0;
    for (int i = 1; i < n; ++i) if (x[i] > x[b]) b = i;
    return b;
}

/* mcts.py:61-68 with temperature 1: N**1.0 is exactly (double)N */
void orc_node_pi(const orc_node* n, double* out) {
    double s = 0.0;
    for (int i = 0; i < n->num; ++i) s += (double)n->N[i];
    if (s > 0) for (int i = 0; i < n->num; ++i) out[i] = (double)n->N[i] / s;
    else for (int i = 0; i < n->num; ++i) out[i] = 1.0 / (double)n->num;
}

int orc_node_greedy_action(const orc_node* n) { /* mcts.py:73-74 */
    double pi[ORC_MAX_ACTIONS];
    orc_node_pi(n, pi);
    return argmax_first_d(pi, n->num);
}

/* mcts.py:70-71 -> legacy numpy RandomState.choice(p=...): cdf = cumsum(p);
 * cdf /= cdf[-1]; index = searchsorted(cdf, u, side='right') */
int orc_node_sample_action(const orc_node* n, double u) {
    double pi[ORC_MAX_ACTIONS], cdf[ORC_MAX_ACTIONS];
    orc_node_pi(n, pi);
    double acc = 0.0;
    for (int i = 0; i < n->num; ++i) { acc += pi[i]; cdf[i] = acc; }
    double last = cdf[n->num - 1];
    int idx = 0;
    for (int i = 0; i < n->num; ++i) { cdf[i] /= last; if (cdf[i] <= u) idx = i + 1; }
    return idx;
}

/* mcts.py:88-91: (1-eps) is a Python float, so (1-eps)*P stays float32; eps*eta
 * is float64; the sum is float64 and is rounded once when stored back */
void orc_node_add_noise(orc_node* n, const double* eta, double epsilon) {
    float one_minus = (float)(1.0 - epsilon);
    for (int i = 0; i < n->num; ++i) {
        float a = one_minus * n->P[i];
        n->P[i] = (float)((double)a + epsilon * eta[i]);
    }
}

void orc_node_add_value_observation(orc_node* n, int ai, float v) { /* mcts.py:84-86 */
    n->W[ai] += v;
    n->N[ai] += 1;
}

long orc_node_count(const orc_node* n) {
    if (!n || !n->children) return 0;
    long c = 1;
    for (int i = 0; i < n->num; ++i) c += orc_node_count(n->children[i]);
    return c;
}

/* ===================================================================== */
/* leaf evaluators                                                        */
/* ===================================================================== */
/* models.py:90-95 through mcts.py:94-107: softmax of equal logits */
void orc_eval_uniform(void* ctx, int game, const uint8_t* hist, int len,
                      const int* valid, int nvalid, float* p_out, float* v_out) {
    (void)ctx; (void)game; (void)hist; (void)len; (void)valid;
    for (int i = 0; i < nvalid; ++i) p_out[i] = 1.0f / (float)nvalid;
    *v_out = 0.0f;
}

/* hash of the rendered planes: sum over set cells i of (i*i*31 + 7*i + 1) */
uint32_t orc_position_hash(int game, const uint8_t* hist, int len) {
    float board[2 * C4_SPACES];
    int h, w;
    orc_board_shape(game, &h, &w);
    orc_render(game, hist, len, board);
    uint32_t acc = 0;
    int n = 2 * h * w;
    for (int i = 0; i < n; ++i)
        if (board[i] != 0.0f) acc += (uint32_t)(i * i * 31 + 7 * i + 1);
    return acc * 2654435761u;
}

void orc_eval_subset(void* ctx, int game, const uint8_t* hist, int len,
                     const int* valid, int nvalid, float* p_out, float* v_out) {
    (void)ctx;
    uint32_t h = orc_position_hash(game, hist, len);
    int k = 0;
    for (int i = 0; i < nvalid; ++i) k += (int)((h >> (8 + valid[i])) & 1u);
    if (k == 0) {
        for (int i = 0; i < nvalid; ++i) p_out[i] = 1.0f / (float)nvalid;
    } else {
        for (int i = 0; i < nvalid; ++i)
            p_out[i] = ((h >> (8 + valid[i])) & 1u) ? 1.0f / (float)k : 0.0f;
    }
    *v_out = (float)((int)(h % 17u) - 8) / 8.0f;
}

/* ===================================================================== */
/* RNG provider for timing runs (statistically, not bitwise, equivalent to */
/* the reference's global numpy RandomState)                              */
/* ===================================================================== */
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix(uint64_t* s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
void orc_rng_seed(orc_rng* r, uint64_t seed) {
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix(&seed);
}
static uint64_t rng_next(orc_rng* r) {
    uint64_t* s = r->s;
    uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
double orc_rng_uniform(void* rng) { return (double)(rng_next((orc_rng*)rng) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_normal(orc_rng* r) {
    double u1, u2;
    do { u1 = orc_rng_uniform(r); } while (u1 <= 0.0);
    u2 = orc_rng_uniform(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static double rng_gamma(orc_rng* r, double a) { /* Marsaglia-Tsang, boost for a < 1 */
    if (a < 1.0) {
        double u;
        do { u = orc_rng_uniform(r); } while (u <= 0.0);
        return rng_gamma(r, a + 1.0) * pow(u, 1.0 / a);
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = rng_normal(r); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = orc_rng_uniform(r);
        if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
        if (u > 0.0 && log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
    }
}
void orc_rng_dirichlet(void* rng, double alpha, int n, double* eta) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) { eta[i] = rng_gamma((orc_rng*)rng, alpha); s += eta[i]; }
    for (int i = 0; i < n; ++i) eta[i] /= s;
}

/* ===================================================================== */
/* run_mcts: mcts.py:110-158                                              */
/* ===================================================================== */
void orc_run_mcts(const orc_config* cfg, const uint8_t* hist, int len, orc_node* root,
                  orc_eval_fn eval, void* eval_ctx,
                  orc_dirichlet_fn dir, void* dir_ctx, orc_stats* stats) {
    const int game = cfg->game;
    int valid[ORC_MAX_ACTIONS];
    float p[ORC_MAX_ACTIONS], v;

    /* mcts.py:114-116: expand the root if needed; its value is discarded */
    if (!orc_node_expanded(root)) {
        int nv = orc_valid_actions(game, hist, len, valid);
        eval(eval_ctx, game, hist, len, valid, nv, p, &v);
        orc_node_expand(root, p, nv);
        if (stats) stats->root_evals++;
    }
    /* mcts.py:117: noise on every call */
    if (dir) {
        double eta[ORC_MAX_ACTIONS];
        dir(dir_ctx, cfg->alpha, root->num, eta);
        orc_node_add_noise(root, eta, cfg->epsilon);
    }

    uint8_t sim[ORC_MAX_MOVES + 1];
    for (int s = 0; s < cfg->num_simulations; ++s) { /* mcts.py:119 */
        orc_node* node = root;
        memcpy(sim, hist, (size_t)len);   /* game.clone(), game.py:35-36 */
        int slen = len;

        /* mcts.py:123-127 */
        while (orc_node_expanded(node) && !orc_terminal(game, sim, slen)) {
            double score[ORC_MAX_ACTIONS];
            orc_node_ucb_score(node, cfg->c_base, cfg->c_init, cfg->lut, cfg->lut_len, score);
            int ai = argmax_first_d(score, node->num);
            node = orc_node_traverse(node, ai);
            orc_valid_actions(game, sim, slen, valid);
            sim[slen++] = (uint8_t)valid[ai];
            if (stats) stats->select_steps++;
        }

        if (orc_terminal(game, sim, slen)) {
            /* mcts.py:131-134 */
            v = (float)orc_terminal_value(game, sim, slen, prev_player(slen));
        } else {
            /* mcts.py:135-144 */
            int nv = orc_valid_actions(game, sim, slen, valid);
            eval(eval_ctx, game, sim, slen, valid, nv, p, &v);
            orc_node_expand(node, p, nv);
            v *= -1.0f;
            if (stats) stats->leaf_evals++;
        }

        /* mcts.py:147-158 */
        while (node->parent != NULL) {
            if (node == root) break;
            orc_node* par = node->parent;
            int ai = 0;
            while (par->children[ai] != node) ai++;   /* children.index(node) */
            orc_node_add_value_observation(par, ai, v);
            v *= -1.0f;
            node = par;
        }
        if (stats) stats->sims++;
    }
}

/* ===================================================================== */
/* SelfPlayWorker.play_game: selfplay.py:12-42                            */
/* ===================================================================== */
void orc_play_game(const orc_config* cfg,
                   orc_eval_fn eval, void* eval_ctx,
                   orc_dirichlet_fn dir, void* dir_ctx,
                   orc_uniform_fn uni, void* uni_ctx,
                   orc_game_record* rec, orc_stats* stats) {
    const int game = cfg->game;
    uint8_t hist[ORC_MAX_MOVES + 1];
    int len = 0;
    orc_node* top = orc_node_new(NULL); /* owns the whole tree for freeing */
    orc_node* tree = top;
    memset(rec, 0, sizeof(*rec));

    while (!orc_terminal(game, hist, len)) {
        orc_run_mcts(cfg, hist, len, tree, eval, eval_ctx, dir, dir_ctx, stats);
        int ai;
        if (len < cfg->exploration_depth) ai = orc_node_sample_action(tree, uni(uni_ctx));
        else ai = orc_node_greedy_action(tree);
        int valid[ORC_MAX_ACTIONS];
        orc_valid_actions(game, hist, len, valid);

        rec->num_edges[len] = (uint8_t)tree->num;
        rec->action_index[len] = (uint8_t)ai;
        for (int i = 0; i < tree->num; ++i) {
            rec->visits[len][i] = tree->N[i];
            rec->root_W[len][i] = tree->W[i];
            rec->root_P[len][i] = tree->P[i];
        }
        rec->history[len] = (uint8_t)valid[ai];
        hist[len++] = (uint8_t)valid[ai];

        if (cfg->reuse_search_tree) {
            tree = orc_node_traverse(tree, ai);
            orc_node_kill_siblings(tree);
        } else {
            orc_node_free(top);
            top = orc_node_new(NULL);
            tree = top;
        }
    }
    rec->num_moves = len;
    rec->winner = orc_winner(game, hist, len);
    orc_node_free(top);
}
