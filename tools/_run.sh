#!/bin/bash
timeout 600 python tools/lat1.py
