from .connectfour import ConnectFour
from .tictactoe import TicTacToe

BY_ID = {TicTacToe.AZ_GAME_ID: TicTacToe, ConnectFour.AZ_GAME_ID: ConnectFour}
BY_NAME = {'ttt': TicTacToe, 'tictactoe': TicTacToe, 'c4': ConnectFour, 'connectfour': ConnectFour}
