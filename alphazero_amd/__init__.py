"""alphazero_amd: MI355X-native self-play hot path (games -> PUCT MCTS -> leaf eval).

Host-side mirror of the reference's Game / MCTreeNode / run_mcts / SelfPlayWorker
surface over the HIP engine in libazhip.so (see include/azhip.h)."""
__all__ = ['game', 'games', 'mcts', 'models', 'selfplay', 'engine', 'config']
