RANDOMSEED = 12345  # global seed used by main.py (reference: change_randomseed.py:1)
