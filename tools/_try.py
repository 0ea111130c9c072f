import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import gpu_checks as gc
for a in [(2, 1728, 27, 8, 32), (1, 216, 27, 10, 32), (2, 13824, 27, 4, 32), (2, 100, 27, 1, 32), (1, 512, 8, 2, 16), (2, 61, 8, 5, 16), (1, 8, 8, 4, 16)]:
    print(gc.check_battn(*a))
