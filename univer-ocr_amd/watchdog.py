"""Progress watchdog for multi-rank runs (bench.py, my_model/train.py).

A data-parallel run can only hang in a collective or a rendezvous (a rank that died, ranks issuing different numbers of
collectives); a hung run burns the launcher's whole timeout and says nothing.  The main thread calls beat() after every
step / phase; when nothing beats for `limit` seconds the process says where it was and exits with code 3.  It never
re-executes anything: the GPU is initialised."""
import os
import sys
import threading
import time


class Watchdog(threading.Thread):
    def __init__(self, limit, rank, tag='watchdog'):
        super().__init__(daemon=True)
        self.limit, self.rank, self.tag = limit, rank, tag
        self.last, self.where = time.monotonic(), 'start'
        self.start()

    def beat(self, where):
        self.last, self.where = time.monotonic(), where

    def run(self):
        while True:
            time.sleep(1.0)
            idle = time.monotonic() - self.last
            if idle > self.limit:
                print(f'[{self.tag}] rank {self.rank}: no progress for {idle:.0f} s in "{self.where}" '
                      f'(limit {self.limit:.0f} s): giving up', file=sys.stderr, flush=True)
                os._exit(3)
