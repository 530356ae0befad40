def main():
    print("ppo golden: not yet")
