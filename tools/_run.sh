#!/bin/bash
timeout 300 python tools/_stem_ab.py
